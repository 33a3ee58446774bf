/*
 * vkmr_hip_experiments.h -- entry points of the EXPERIMENTS build only (build/ab/libexp.so, -DVKMR_EXPERIMENTS): measured,
 * kept for the record, not part of the product library or of the drop-in boundary (include/vkmr_hip.h).
 */
#ifndef VKMR_HIP_EXPERIMENTS_H
#define VKMR_HIP_EXPERIMENTS_H

#include "vkmr_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/*
 * Round 3 shipped this as an opt-in mode of the front end (VKMR_DEVICE_SPLIT=1: spans of text cross PCIe as they are and are
 * split on the device).  It printed the same 60-70 ms as the host's two-pass packer (profiles/r03_frontend_device_split.txt):
 * what costs is reading 2 GB of mapped input and writing 2 GB of pinned memory, which both paths do.  An opt-in mode that
 * measured no gain is not a product feature: kernels, entry points and the host path live in the experiments
 * build since round 4.
 */
/*
 * TEXT -> PACKED BATCH, on the device.  Newline-separated text that already lies in device memory becomes what the
 * reference's input loop and Batch::Push build on the host one string at a time: every non-empty line is a string
 * (Input::Get, src/vkmr/Inputs.cpp:75-101: a line ends at '\n', '\r' is kept; run() skips empty lines,
 * src/vkmr/Vkmr.cpp:38-51), the strings lie back to back on word boundaries with zero padding from word 0 of data_dev,
 * and meta_dev[i] = {start word, size} (src/vkmr/Batches.cpp:64-121).  vkmr_hip_map_async takes the result as it is.
 *   text_dev     text_bytes bytes (below 4 GiB), 16-byte aligned, readable up to the next multiple of 16 plus 16; the
 *                text must END IN '\n' (append one to a last line that has none): what follows the last '\n' is ignored
 *   scratch_dev  vkmr_hip_split_scratch_bytes(text_bytes, meta_capacity) bytes
 *   result_dev   three words, written: [0] strings found, [1] words they take, [2] 0 -- or 1 when they do not fit
 *                meta_capacity entries / data_capacity_words words (then what was written is not to be used)
 */
VKMR_API vkmr_status vkmr_hip_split_text_async(int dev, vkmr_stream s, const uint8_t* text_dev, uint32_t text_bytes, void* scratch_dev,
                                               uint32_t* data_dev, uint64_t data_capacity_words, vkmr_metadata* meta_dev,
                                               uint32_t meta_capacity, uint32_t* result_dev);
VKMR_API size_t vkmr_hip_split_scratch_bytes(uint32_t text_bytes, uint32_t meta_capacity);

#ifdef __cplusplus
}
#endif

#endif
