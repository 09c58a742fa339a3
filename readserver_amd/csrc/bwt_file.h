// bwt_file.h -- SGA run-length BWT files (the ".bwt" ReadServer loads).
// Layout read by BWTReaderRLE (src/bwt/rlebwt_reader.cpp:27-48, include/bwt/rlebwt_reader.h:17-23):
//   u16 magic 0xCACA | u64 num_strings | u64 num_symbols | u64 num_runs | u32 flag (0) | run bytes
// little endian, 30 header bytes.
#ifndef RSBWT_BWT_FILE_H
#define RSBWT_BWT_FILE_H

#include <stdint.h>
#include <stdio.h>

namespace rsb {

constexpr uint16_t RLBWT_MAGIC = 0xCACA;
constexpr size_t RLBWT_HEADER_BYTES = 30;

struct bwt_header {
    uint64_t num_strings;
    uint64_t num_symbols;
    uint64_t num_runs;
    uint32_t flag;
};

// 0 ok, otherwise an RSBWT_E* code.  On success *f is positioned at the first run byte.
int bwt_open_read(const char *path, FILE **f, bwt_header *hdr);
int bwt_write(const char *path, const bwt_header &hdr, const uint8_t *runs);

}  // namespace rsb
#endif
