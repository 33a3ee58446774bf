// stream_pack.hpp -- splitting a byte stream into lines with the reference's rules
// and packing them into the batch layout.
#pragma once
#include <cstddef>
#include <cstdint>

#include "vkmr_hip.h"

namespace vkmr {

// Words needed for a string of `len` bytes (Batch::WordCount, reference
// src/vkmr/Batches.cpp:182-187).
inline uint32_t WordCount(size_t len) { return (uint32_t)((len + 3u) / 4u); }

// Result of packing a span of text.
struct PackResult {
    uint64_t consumed;   // bytes of input consumed (always ends just after a '\n', or at len when `final`)
    uint64_t strings;    // strings appended
    uint64_t words;      // words appended
    uint64_t bytes;      // payload bytes appended (sum of sizes)
    uint64_t empties;    // empty lines skipped
};

// Appends the non-empty lines of buf[0,len) to data/meta.  A line ends at '\n' or,
// when `final` is set, at the end of the buffer; '\r' is kept; empty lines are never
// strings (Input::Get, reference src/vkmr/Inputs.cpp:75-101; run(), src/vkmr/Vkmr.cpp:38-51).
// Stops early when either buffer is full.  `first_word` is the word index at which
// the first new string is placed.
PackResult PackLines(const uint8_t* buf, size_t len, bool final, uint32_t* data, uint64_t first_word,
                     uint64_t data_capacity_words, vkmr_metadata* meta, uint64_t meta_capacity);

// What PackLines would append for buf[0,len) with final = true, without writing anything
// (first pass of the parallel packer).
struct LineCount { uint64_t strings, words, bytes, empties; bool too_long; uint32_t longest = 0; };   // longest: IndexLines only
LineCount CountLines(const uint8_t* buf, size_t len);

// Indexed two-pass form, what the parallel packer runs on each part of a span (Batch::PushLinesParallel): pass 1 records
// where every line of the part ends and returns what the part will append -- the prefix sums over the parts give every
// part its place in the batch -- and pass 2 packs from that record without looking for newlines again.
//   IndexLines   ends[i] = offset of the i-th '\n' of buf[0,len); an unterminated last line ends at len.  len < 2^32 - 64.
//   PackIndexed  appends the non-empty lines at data[first_word ...) / meta[0 ...), and their sizes as 16-bit numbers
//                (saturated at 65 535) at sizes[0 ...) when that is not null; writes nothing at or beyond
//                data[end_word] (the next part's words are another thread's) and reads nothing beyond buf[len).
// Same words and metadata as PackLines(final = true) on the same bytes (tests/test_host_tools.py, test_host_fuzz.py).
struct LineIndex {
    uint32_t* ends = nullptr;
    size_t cap = 0, count = 0;
    LineIndex() = default;
    LineIndex(const LineIndex&) = delete;
    LineIndex& operator=(const LineIndex&) = delete;
    LineIndex(LineIndex&& o) noexcept : ends(o.ends), cap(o.cap), count(o.count) { o.ends = nullptr; o.cap = o.count = 0; }
    ~LineIndex();
    void Reserve(size_t lines);   // room for `lines` entries plus the slack the vector form writes past the end; contents are dropped
};
LineCount IndexLines(const uint8_t* buf, size_t len, LineIndex* ix);
// streaming: the packed words go out with non-temporal stores (same words; which is faster depends on how busy the host's
// memory is: Batches' PackTuner tries both).
void PackIndexed(const uint8_t* buf, size_t len, const LineIndex& ix, uint32_t* data, uint64_t first_word, uint64_t end_word,
                 vkmr_metadata* meta, uint16_t* sizes = nullptr, bool streaming = false);
LineCount IndexLinesPortable(const uint8_t* buf, size_t len, LineIndex* ix);
void PackIndexedPortable(const uint8_t* buf, size_t len, const LineIndex& ix, uint32_t* data, uint64_t first_word, uint64_t end_word,
                         vkmr_metadata* meta, uint16_t* sizes = nullptr);

// One pass for the device-side splitter (vkmr_hip_split_text_async): buf[0,len) is copied to dst as it is, and its lines
// are counted -- newlines, and those among them that end an EMPTY line (the byte before them is a newline too;
// `after_newline` says whether the byte before buf[0] was one, true at the start of a stream).  Strings = newlines - empties.
struct TextCount { uint64_t newlines, empties; };
TextCount CopyAndCountLines(const uint8_t* buf, size_t len, uint8_t* dst, bool after_newline);
TextCount CopyAndCountLinesPortable(const uint8_t* buf, size_t len, uint8_t* dst, bool after_newline);

// The portable forms (one memchr per line).  PackLines / CountLines use AVX2 forms where the CPU has them (the newline
// positions of 64 input bytes at a time); these stay as the reference the tests compare them with.
PackResult PackLinesPortable(const uint8_t* buf, size_t len, bool final, uint32_t* data, uint64_t first_word,
                             uint64_t data_capacity_words, vkmr_metadata* meta, uint64_t meta_capacity);
LineCount CountLinesPortable(const uint8_t* buf, size_t len);

}  // namespace vkmr
