// bwt_file.cpp -- see bwt_file.h
#include "bwt_file.h"

#include <string.h>
#include <sys/stat.h>

#include "../../include/rsbwt.h"

namespace rsb {

int bwt_open_read(const char *path, FILE **fout, bwt_header *hdr) {
    FILE *f = fopen(path, "rb");
    if (!f) return RSBWT_EIO;
    uint8_t h[RLBWT_HEADER_BYTES];
    if (fread(h, 1, sizeof h, f) != sizeof h) { fclose(f); return RSBWT_EFORMAT; }
    uint16_t magic;
    memcpy(&magic, h, 2);
    if (magic != RLBWT_MAGIC) { fclose(f); return RSBWT_EFORMAT; }
    memcpy(&hdr->num_strings, h + 2, 8);
    memcpy(&hdr->num_symbols, h + 10, 8);
    memcpy(&hdr->num_runs, h + 18, 8);
    memcpy(&hdr->flag, h + 26, 4);
    struct stat st;
    if (fstat(fileno(f), &st) == 0 && S_ISREG(st.st_mode) &&
        (uint64_t)st.st_size < RLBWT_HEADER_BYTES + hdr->num_runs) {
        fclose(f);
        return RSBWT_EFORMAT;  // truncated
    }
    *fout = f;
    return RSBWT_OK;
}

int bwt_write(const char *path, const bwt_header &hdr, const uint8_t *runs) {
    FILE *f = fopen(path, "wb");
    if (!f) return RSBWT_EIO;
    uint8_t h[RLBWT_HEADER_BYTES];
    const uint16_t magic = RLBWT_MAGIC;
    memcpy(h, &magic, 2);
    memcpy(h + 2, &hdr.num_strings, 8);
    memcpy(h + 10, &hdr.num_symbols, 8);
    memcpy(h + 18, &hdr.num_runs, 8);
    memcpy(h + 26, &hdr.flag, 4);
    int rc = RSBWT_OK;
    if (fwrite(h, 1, sizeof h, f) != sizeof h) rc = RSBWT_EIO;
    if (rc == RSBWT_OK && hdr.num_runs && fwrite(runs, 1, hdr.num_runs, f) != hdr.num_runs)
        rc = RSBWT_EIO;
    if (fclose(f) != 0) rc = RSBWT_EIO;
    return rc;
}

}  // namespace rsb
