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

// The raw 64-bit outputs in BULK: a 128-bit LCG step costs a dependent 128 x 128 multiply, and a sweep at n = 100 000 consumes
// ~230 000 of them one after the other -- two thirds of the 0.74 ms a plan took (profiles/r04a_profile_c5.txt: the host thread that
// replays the stream, not the device, set the pace of configs[4]'s compound step).  The outputs do not depend on what is done with
// them, so they are produced a block ahead by FOUR interleaved lanes of the same sequence, lane j holding every fourth state:
// S_{k+4} = a^4 S_k + c (a^3 + a^2 + a + 1) -- four independent multiply chains the core overlaps -- and the consumers below
// (`next32` buffering, `random_interval`, Lemire, `next_double`) read them from the block.  The generator state handed back is
// S_{consumed}: the initial state advanced by the number of outputs actually used (`advance`, the O(log n) jump of an LCG).
struct Pcg64Replay {
  unsigned __int128 state, inc;
  int has_uint32;
  uint32_t uinteger;
  // bulk mode (begin_bulk): outputs of the states after `bstate`, consumed in order
  static constexpr int BLK = 2048;
  bool bulk = false;
  unsigned __int128 bstate = 0, state0 = 0;   // state at the start of the current block / when bulk mode began
  uint64_t consumed = 0;                      // outputs handed out since begin_bulk
  int bpos = BLK;
  uint64_t buf[BLK];

  static unsigned __int128 mult() {
    return ((unsigned __int128)0x2360ED051FC65DA4ull << 64) | 0x4385DF649FCCF645ull;   // PCG_DEFAULT_MULTIPLIER_128
  }
  static uint64_t output(unsigned __int128 st) {   // XSL-RR 128/64
    const uint64_t hi = (uint64_t)(st >> 64), lo = (uint64_t)st;
    const unsigned rot = (unsigned)(st >> 122);
    const uint64_t x = hi ^ lo;
    return (x >> rot) | (x << ((-rot) & 63));
  }
  // state after `delta` steps (pcg_advance_lcg_128)
  static unsigned __int128 advance(unsigned __int128 st, unsigned __int128 inc_, uint64_t delta) {
    unsigned __int128 acc_mult = 1, acc_plus = 0, cur_mult = mult(), cur_plus = inc_;
    while (delta > 0) {
      if (delta & 1) { acc_mult *= cur_mult; acc_plus = acc_plus * cur_mult + cur_plus; }
      cur_plus = (cur_mult + 1) * cur_plus;
      cur_mult *= cur_mult;
      delta >>= 1;
    }
    return acc_mult * st + acc_plus;
  }
  void begin_bulk() { bulk = true; bstate = state0 = state; consumed = 0; bpos = BLK; }
  void end_bulk() { state = advance(state0, inc, consumed); bulk = false; }
  void refill() {
    const unsigned __int128 a = mult(), a2 = a * a, a4 = a2 * a2, c4 = inc * (a2 * a + a2 + a + 1);
    unsigned __int128 l0 = bstate * a + inc, l1 = l0 * a + inc, l2 = l1 * a + inc, l3 = l2 * a + inc;
    for (int i = 0; i < BLK; i += 4) {
      buf[i] = output(l0); buf[i + 1] = output(l1); buf[i + 2] = output(l2); buf[i + 3] = output(l3);
      if (i + 4 == BLK) bstate = l3;
      l0 = l0 * a4 + c4; l1 = l1 * a4 + c4; l2 = l2 * a4 + c4; l3 = l3 * a4 + c4;
    }
    bpos = 0;
  }
  uint64_t next64() {
    if (bulk) {
      if (bpos == BLK) refill();
      ++consumed;
      return buf[bpos++];
    }
    state = state * mult() + inc;
    return output(state);
  }
  // The generator after n elements of the per-element loop (`choice(k - 1)` + `uniform()`, every dimension with the SAME k), assuming
  // no Lemire rejection (probability (2^32 mod (k-1)) / 2^32 per draw: zero for k - 1 a power of two; the caller checks what the
  // real pass reports): element t takes a 32-bit half (the buffered one, or the low half of a fresh output whose high half is then
  // buffered) and one whole output for the double -- so the number of outputs consumed, the buffered flag and the buffered half
  // follow from n and the flag alone.  Lets the NEXT sweep's shuffle start while this sweep's draws are still being replayed.
  void skip_draws(uint64_t n, uint32_t k) {
    if (n == 0) return;
    if (k <= 2) { state = advance(state, inc, n); return; }   // choice(1) draws nothing
    const uint64_t need = n - (has_uint32 ? 1 : 0);            // halves that must come from fresh outputs
    const uint64_t refills = (need + 1) / 2;
    if (refills > 0) {
      // the last refill happens at element t_last = (has_uint32 ? 1 : 0) + 2 (refills - 1); before it: t_last doubles + refills - 1 refills
      const uint64_t t_last = (has_uint32 ? 1 : 0) + 2 * (refills - 1);
      const uint64_t pos = t_last + (refills - 1);            // outputs consumed before that refill
      const uint64_t o = output(advance(state, inc, pos + 1));
      uinteger = (uint32_t)(o >> 32);
      has_uint32 = (need & 1) ? 1 : 0;                        // an odd number of fresh halves leaves the last high half buffered
    } else has_uint32 = 0;                                     // (n == 1 and the buffered half was used)
    state = advance(state, inc, n + refills);
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
  // `Generator.shuffle` of a list / 1-d array of n entries (Fisher-Yates from the top: j = random_interval(i) for i = n-1 .. 1,
  // swap(order[i], order[j])) in two passes.  The masked-rejection loop of `random_interval` is a data-dependent branch that
  // fails 0-50 % of the time -- mispredicted every third element or so, which (with the swap's dependent loads behind it) made the
  // shuffle the longest stage of a plan.  Here the stream of 32-bit halves is walked WITHOUT a branch on acceptance: the
  // candidate is stored to js[i] whatever it is and i only steps down when it was accepted (`i -= v <= i`); the mask is constant
  // while i stays between two powers of two.  The swaps then run over known indices (independent loads the core overlaps).
  // Exactly the halves `interval()` would consume are consumed, in the same order (buffered half first, then low / high of each
  // output).  n - 1 <= 2^32 - 1 (a 32-bit interval).
  template <typename T>
  void shuffle(int64_t n, T* order, int32_t* js /* scratch, n entries */) {
    if (n < 2) return;
    int64_t i = n - 1;
    uint64_t half_buf[BLK];
    const uint32_t* hp = nullptr;
    const uint32_t* hend = nullptr;
    bool first = has_uint32 != 0;
    uint32_t pend = uinteger;
    uint64_t fresh_halves = 0;       // halves taken from fresh outputs
    unsigned __int128 st = state;
    const unsigned __int128 a = mult(), a2 = a * a, a4 = a2 * a2, c4 = inc * (a2 * a + a2 + a + 1);
    uint32_t last_hi = 0;
    while (i >= 1) {
      uint64_t mask = (uint64_t)i;
      mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
      const int64_t lo = (int64_t)(mask >> 1) + 1;
      if (first) {                  // the half NumPy had buffered before the shuffle began
        first = false;
        const int64_t v = (int64_t)(pend & mask);
        js[i] = (int32_t)v;
        i -= v <= i;
        continue;
      }
      if (hp == hend) {             // the next BLK outputs: four interleaved lanes of the LCG
        unsigned __int128 l0 = st * a + inc, l1 = l0 * a + inc, l2 = l1 * a + inc, l3 = l2 * a + inc;
        for (int q = 0; q < BLK; q += 4) {
          half_buf[q] = output(l0); half_buf[q + 1] = output(l1); half_buf[q + 2] = output(l2); half_buf[q + 3] = output(l3);
          if (q + 4 == BLK) st = l3;
          l0 = l0 * a4 + c4; l1 = l1 * a4 + c4; l2 = l2 * a4 + c4; l3 = l3 * a4 + c4;
        }
        hp = reinterpret_cast<const uint32_t*>(half_buf);    // (little-endian: low half, then high half -- next32's order)
        hend = hp + 2 * BLK;
      }
      const uint32_t* h = hp;
      const uint32_t m32 = (uint32_t)mask;
      while (h < hend && i >= lo) {
        const int64_t v = (int64_t)(*h++ & m32);
        js[i] = (int32_t)v;
        i -= v <= i;
      }
      fresh_halves += (uint64_t)(h - hp);
      if (h > hp) last_hi = h[-1];
      hp = h;
    }
    // the generator as `interval()` calls would have left it
    const uint64_t outs = (fresh_halves + 1) / 2;
    if (fresh_halves > 0) {
      if (fresh_halves & 1) {        // the low half of the last output was the last one used: its high half stays buffered
        has_uint32 = 1;
        uinteger = hp[0];            // (the half that follows the last one consumed, inside the same output)
      } else { has_uint32 = 0; uinteger = last_hi; }   // (NumPy keeps the used half in `uinteger`)
    } else has_uint32 = 0;           // only the buffered half was needed
    state = advance(state, inc, outs);
    for (int64_t q = n - 1; q >= 1; --q) {
      const T t = order[q];
      const int32_t j = js[q];
      order[q] = order[j];
      order[j] = t;
    }
  }
  // The per-element loop for n elements that all have the same k > 2 (`choice(k - 1)` then `uniform()`), written against the raw
  // output stream directly: a 32-bit half for Lemire's draw (the buffered one, or the low half of a fresh output whose high half
  // is buffered for the next element), then one whole output for the double.  Returns the number of elements done: n, or the
  // index of the first element whose Lemire draw needs the rejection loop (probability < k / 2^32 per element) -- the caller
  // continues from there with `integers()` / `next_double()`, the generator standing exactly before that element.
  int64_t draws_same_k(int64_t n, uint32_t k, int32_t* cand, double* uniform) {
    const uint32_t rng_excl = k - 1;
    uint64_t raw[BLK];
    int pos = BLK;
    unsigned __int128 st = state;
    const unsigned __int128 a = mult(), a2 = a * a, a4 = a2 * a2, c4 = inc * (a2 * a + a2 + a + 1);
    uint64_t used = 0;
    auto refill_raw = [&]() {
      unsigned __int128 l0 = st * a + inc, l1 = l0 * a + inc, l2 = l1 * a + inc, l3 = l2 * a + inc;
      for (int q = 0; q < BLK; q += 4) {
        raw[q] = output(l0); raw[q + 1] = output(l1); raw[q + 2] = output(l2); raw[q + 3] = output(l3);
        if (q + 4 == BLK) st = l3;
        l0 = l0 * a4 + c4; l1 = l1 * a4 + c4; l2 = l2 * a4 + c4; l3 = l3 * a4 + c4;
      }
      pos = 0;
    };
    int have = has_uint32;
    uint32_t pend = uinteger;
    int64_t t = 0;
    for (; t < n; ++t) {
      // (state before this element, should the slow path have to take over)
      const int have0 = have; const uint32_t pend0 = pend; const uint64_t used0 = used;
      uint32_t h;
      if (have) { h = pend; have = 0; }
      else {
        if (pos == BLK) refill_raw();
        const uint64_t o = raw[pos++]; ++used;
        h = (uint32_t)o; pend = (uint32_t)(o >> 32); have = 1;
      }
      const uint64_t m = (uint64_t)h * rng_excl;
      if ((uint32_t)m < rng_excl) {     // Lemire's test may reject: hand this element back untouched
        have = have0; pend = pend0; used = used0;
        break;
      }
      cand[t] = (int32_t)(m >> 32);
      if (pos == BLK) refill_raw();
      uniform[t] = (double)(raw[pos++] >> 11) * (1.0 / 9007199254740992.0); ++used;
    }
    has_uint32 = have; uinteger = pend;
    state = advance(state, inc, used);
    return t;
  }
  int rejections = 0;   // Lemire rejections since the object was set up (skip_draws assumes there are none)
  uint32_t lemire32(uint32_t rng) {   // buffered_bounded_lemire_uint32: uniform on [0, rng]
    const uint32_t rng_excl = rng + 1;
    uint64_t m = (uint64_t)next32() * rng_excl;
    uint32_t leftover = (uint32_t)(m & 0xffffffffu);
    if (leftover < rng_excl) {
      const uint32_t threshold = (0xffffffffu - rng) % rng_excl;
      while (leftover < threshold) {
        ++rejections;
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
