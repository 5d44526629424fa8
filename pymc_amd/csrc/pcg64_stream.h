// Host-side replay of what one sweep of the reference's CategoricalGibbsMetropolis draws from `step.rng`
// (pymc/step_methods/metropolis.py:771-786, 1225-1229; pymc/step_methods/arraystep.py:208-235):
//
//   rng.shuffle(dimcats)                       one `random_interval(i)` per i = n-1 .. 1 (Fisher-Yates on a Python list)
//   per element, in the shuffled order:
//     rng.choice(k - 1)                        = rng.integers(0, k - 1): Lemire's bounded draw on BUFFERED 32-bit halves
//     rng.uniform()                            one 64-bit draw -> 53-bit double (the caller takes NumPy's log of it)
//
// The per-element proposals are conditionally independent, so the device evaluates all of them at once -- but each element
// must see exactly the random numbers the sequential loop would have handed it.  NumPy's Generator cannot produce that
// interleaved stream in bulk (the 32-bit draws consume cached halves of 64-bit outputs between the 64-bit draws), and 2 n
// Python-level calls per sweep would cost seconds at n = 100 000; so the stream is replayed here from the generator's
// state: PCG64 (XSL-RR 128/64, the default `BitGenerator`), NumPy's `next_uint32` buffering, `random_interval` (masked
// rejection), `buffered_bounded_lemire_uint32`, and `next_double`.  Checked bit for bit against NumPy in
// tests/test_gibbs.py.
#pragma once
#include <cmath>
#include <cstdint>

struct Pcg64Replay {
  unsigned __int128 state, inc;
  int has_uint32;
  uint32_t uinteger;

  static unsigned __int128 mult() {
    return ((unsigned __int128)0x2360ED051FC65DA4ull << 64) | 0x4385DF649FCCF645ull;   // PCG_DEFAULT_MULTIPLIER_128
  }
  uint64_t next64() {
    state = state * mult() + inc;
    const uint64_t hi = (uint64_t)(state >> 64), lo = (uint64_t)state;
    const unsigned rot = (unsigned)(state >> 122);
    const uint64_t x = hi ^ lo;
    return (x >> rot) | (x << ((-rot) & 63));
  }
  uint32_t next32() {   // NumPy's pcg64_next32: the high half of a 64-bit output is kept for the next call
    if (has_uint32) { has_uint32 = 0; return uinteger; }
    const uint64_t n = next64();
    has_uint32 = 1;
    uinteger = (uint32_t)(n >> 32);
    return (uint32_t)(n & 0xffffffffu);
  }
  double next_double() { return (double)(next64() >> 11) * (1.0 / 9007199254740992.0); }
  uint64_t interval(uint64_t max) {   // random_interval (distributions.c): used by Generator.shuffle
    if (max == 0) return 0;
    uint64_t mask = max, value;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
    if (max <= 0xffffffffull) { while ((value = (next32() & mask)) > max) {} }
    else { while ((value = (next64() & mask)) > max) {} }
    return value;
  }
  uint32_t lemire32(uint32_t rng) {   // buffered_bounded_lemire_uint32: uniform on [0, rng]
    const uint32_t rng_excl = rng + 1;
    uint64_t m = (uint64_t)next32() * rng_excl;
    uint32_t leftover = (uint32_t)(m & 0xffffffffu);
    if (leftover < rng_excl) {
      const uint32_t threshold = (0xffffffffu - rng) % rng_excl;
      while (leftover < threshold) {
        m = (uint64_t)next32() * rng_excl;
        leftover = (uint32_t)(m & 0xffffffffu);
      }
    }
    return (uint32_t)(m >> 32);
  }
  // rng.integers(0, high) for 0 < high <= 2^32 (what rng.choice(high) calls): nothing is drawn when high == 1
  uint32_t integers(uint32_t high) {
    const uint32_t rng = high - 1;
    if (rng == 0) return 0;
    if (rng == 0xffffffffu) return next32();
    return lemire32(rng);
  }
};
