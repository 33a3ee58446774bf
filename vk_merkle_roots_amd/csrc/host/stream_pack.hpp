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
struct LineCount { uint64_t strings, words, bytes, empties; bool too_long; };
LineCount CountLines(const uint8_t* buf, size_t len);

// The portable forms (one memchr per line).  PackLines / CountLines use AVX2 forms where the CPU has them (the newline
// positions of 64 input bytes at a time); these stay as the reference the tests compare them with.
PackResult PackLinesPortable(const uint8_t* buf, size_t len, bool final, uint32_t* data, uint64_t first_word,
                             uint64_t data_capacity_words, vkmr_metadata* meta, uint64_t meta_capacity);
LineCount CountLinesPortable(const uint8_t* buf, size_t len);

}  // namespace vkmr
