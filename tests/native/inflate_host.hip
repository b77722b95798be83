// Test infrastructure: the device inflater's decoder (platypus_amd/csrc/pg_inflate.hip.h) instantiated
// for the host with one lane, so that the CPU tests can run it against zlib without a GPU.
#include <stdlib.h>
#include <string.h>
#include "../../platypus_amd/csrc/pg_inflate.hip.h"

extern "C" int infl_host(const unsigned char* raw, unsigned in_off, unsigned in_len, unsigned char* out, unsigned out_len) {
    unsigned short* tab = static_cast<unsigned short*>(calloc(pg::IT_U16, sizeof(unsigned short)));
    pg::InflTab<1> T;
    T.t = tab;
    pg::InflBlock K;
    K.in_off = in_off; K.in_len = in_len; K.out_off = 0; K.out_len = out_len;
    // the decoder reads its input as aligned words and up to three words past the stream
    unsigned char* in = static_cast<unsigned char*>(aligned_alloc(16, (in_off + in_len + 16 + 15) / 16 * 16 + 16));
    memset(in, 0, (in_off + in_len + 16 + 15) / 16 * 16 + 16);
    memcpy(in, raw, in_off + in_len);
    const int rc = pg::infl_block<1>(T, in, K, out);
    free(in);
    free(tab);
    return rc;
}
