/* Plain-C restatement (native uint32_t arithmetic) of the dropout keep decisions -- TEST INFRASTRUCTURE: cross-checks tests/dropout_ref.py
 * (numpy, 32-bit wraparound emulated in uint64 arrays) on the CPU.  Same specification: csrc/common.h hash_u32 / drop_pair_bits /
 * drop_keep / drop_thresh / eff_seed, csrc/attention.hip attn_row_seed / attn_pair_bits / attn_keep_lo / attn_keep_hi.
 * usage: dropout_ref flat <seed> <p> <first> <count>  |  attn <seed> <p> <Bn> <H> <Sq> <Sk>      -> one '0' / '1' per element */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint32_t umul24(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)(a & 0xffffffu) * (uint64_t)(b & 0xffffffu)) & 0xffffffffu); }
static uint32_t mix24(uint32_t x) {
  x ^= x >> 16; x = umul24(x, 0x85EBCBu);
  x ^= x >> 13; x = umul24(x, 0xC2B2AFu);
  x ^= x >> 16;
  return x;
}
static uint32_t hash_u32(uint64_t seed, uint64_t idx) {
  uint32_t x = (uint32_t)idx ^ (uint32_t)seed;
  x ^= ((uint32_t)(idx >> 32) ^ (uint32_t)(seed >> 32)) * 0x9E3779B9u;
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
static uint32_t drop_pair_bits(uint64_t seed, uint64_t pair) {
  uint32_t x = (uint32_t)pair * 0x9E3779B9u + (uint32_t)seed;
  x += ((uint32_t)(pair >> 32) ^ (uint32_t)(seed >> 32)) * 0x85EBCA6Bu;
  return mix24(x);
}
static uint32_t drop_thresh(float p) {
  double t = (double)p * 4294967296.0;
  if (t < 0) t = 0;
  if (t > 4294967295.0) t = 4294967295.0;
  return (uint32_t)t;
}

int main(int argc, char** argv) {
  if (argc < 6) return 2;
  const uint64_t seed = strtoull(argv[2], 0, 0);
  const uint32_t th = drop_thresh((float)atof(argv[3]));
  if (!strcmp(argv[1], "flat")) {
    const uint64_t first = strtoull(argv[4], 0, 0), count = strtoull(argv[5], 0, 0);
    for (uint64_t i = first; i < first + count; ++i) {
      const uint32_t w = drop_pair_bits(seed, i >> 1);
      putchar((((i & 1) ? (w >> 16) : (w & 0xffffu)) >= (th >> 16)) ? '1' : '0');
    }
  } else {
    if (argc < 8) return 2;
    const int Bn = atoi(argv[4]), H = atoi(argv[5]), Sq = atoi(argv[6]), Sk = atoi(argv[7]);
    const int ts = (int)(th >> 16) - 32768;
    for (int b = 0; b < Bn; ++b) for (int h = 0; h < H; ++h) for (int q = 0; q < Sq; ++q) {
      const uint32_t rs = hash_u32(seed, ((uint64_t)b * H + h) * Sq + q);
      for (int k = 0; k < Sk; ++k) {
        const uint32_t w = mix24(rs + (uint32_t)(k >> 1) * 0x9E3779B9u);
        const int half = (k & 1) ? ((int32_t)w >> 16) : (int)(int16_t)(w & 0xffffu);
        putchar(half >= ts ? '1' : '0');
      }
    }
  }
  putchar('\n');
  return 0;
}
