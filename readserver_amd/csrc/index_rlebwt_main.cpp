// index_rlebwt -- twin of the reference's src/util/index_rlebwt.cpp:7-26: given "<x>.bwt", writes
// the FM-index the CPU reference loads at start-up as "<x>.bwt.bpi2" (byte-identical to
// RLEBWT::serialiseFMIndex).  Host code only: it links none of the GPU engine.
#include <stdio.h>

#include <string>

#include "../../include/rsbwt.h"
#include "bpi2.h"

int main(int argc, char **argv) {
    if (argc != 2) {
        fprintf(stderr, "usage: %s <file.bwt>\n", argv[0]);
        return 1;
    }
    const std::string bwt = argv[1], out = bwt + ".bpi2";
    rsb::bpi2_index ix;
    std::string err;
    int rc = rsb::bpi2_from_bwt(bwt.c_str(), &ix, &err);
    if (rc == RSBWT_OK) rc = rsb::bpi2_save(ix, out.c_str(), &err);
    if (rc != RSBWT_OK) {
        fprintf(stderr, "index_rlebwt: %s\n", err.c_str());
        return 2;
    }
    return 0;
}
