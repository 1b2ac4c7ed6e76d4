// Lists the block starts of a zlib stream as system zlib's inflate(Z_BLOCK) sees them: bit position, octets of output
// before it, BFINAL, BTYPE of the block that starts there.  Used to check K0b's candidates against the truth
// (TBZ_DEBUG_CANDS=<file> makes the engine dump its merged candidate list):
//   gcc -O2 tools/zlib_blocks.c -o /tmp/zlib_blocks -lz && /tmp/zlib_blocks stream.z
// Test / measurement helper; nothing in the product or the test suites depends on it.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
int main(int argc, char** argv) {
  FILE* f = fopen(argv[1], "rb");
  fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
  unsigned char* in = malloc(n); fread(in, 1, n, f); fclose(f);
  z_stream z; memset(&z, 0, sizeof z);
  inflateInit(&z);
  static unsigned char out[1 << 16];
  z.next_in = in; z.avail_in = n;
  unsigned long long outp = 0;
  for (;;) {
    z.next_out = out; z.avail_out = sizeof out;
    int r = inflate(&z, Z_BLOCK);
    outp += sizeof out - z.avail_out;
    if (r != Z_OK && r != Z_STREAM_END) { fprintf(stderr, "err %d\n", r); return 1; }
    if (z.data_type & 128) {  // at a block boundary (or after the header)
      unsigned long long bit = (unsigned long long)(z.next_in - in) * 8 - (z.data_type & 63);
      // type of the block that starts here
      unsigned v = (in[bit >> 3] | (in[(bit >> 3) + 1] << 8)) >> (bit & 7);
      printf("%llu %llu %u %u\n", bit, outp, v & 1, (v >> 1) & 3);
    }
    if (r == Z_STREAM_END) break;
  }
  return 0;
}
