// Philox4x32-10 counter-based RNG (Salmon, Moraes, Dror, Shaw, SC'11), one stream per photon.
//   key     = (seed0, seed1)            -- the driver's seed = (/iseed, batch/) (monteCarloDriver.f95:277)
//   counter = (photon_lo, photon_hi, block, 0)
// Deviates are converted exactly as the reference's getRandomReal does (Code/RandomNumbersForMC.f95:275-299):
//   real( dble(unsigned 32-bit int) / (2**32 - 1) )  in [0, 1], both ends reachable.
// (float)((double)u * (1.0 / 4294967295.0)) equals that quotient for every one of the 2^32 inputs
// (checked exhaustively on the host), and avoids a float64 divide per deviate.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace i3rc {

struct Philox4 { uint32_t v[4]; };

// A round is two 32 x 32 -> 64-bit multiplies (v_mad_u64_u32) and two three-input exclusive ors: gfx950's v_bitop3_b32 (any
// three-input bitwise function, truth table 0x96 = a ^ b ^ c) does `hi ^ counter ^ key` in ONE instruction where the compiler
// emits two v_xor_b32 -- four vector instructions a round instead of six, the same bits.
#ifndef I3RC_PHILOX_ROUNDS
#define I3RC_PHILOX_ROUNDS 10   /* (measurement knob: tools/README.md "Philox rounds"; the tests pin 10) */
#endif
#ifdef I3RC_PHILOX_PLAIN_XOR   /* (measurement knob: the round as the compiler emits it from a ^ b ^ c) */
__device__ inline uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return a ^ b ^ c; }
#else
__device__ inline uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
#endif
__device__ inline Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                        uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < I3RC_PHILOX_ROUNDS; ++r) {
    const uint64_t p0 = (uint64_t)M0 * c0;
    const uint64_t p1 = (uint64_t)M1 * c2;
    const uint32_t n0 = xor3((uint32_t)(p1 >> 32), c1, k0);
    const uint32_t n2 = xor3((uint32_t)(p0 >> 32), c3, k1);
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += W0;
    k1 += W1;
  }
  Philox4 o;
  o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
  return o;
}

// (measurement knob I3RC_UNIT_FLOAT, round 5 -- DESIGN.md section 8: 1 = the same float for every one of the 2^32 inputs without the
// float64 multiply -- the quotient u / (2^32 - 1) = u (1 + 2^-32 + ...) lies strictly between u and u + 1, so it rounds as u does
// except on a tie, which it breaks upwards: a sticky bit in the float64's last place says so to the float32 conversion, and the
// scaling by 2^-32 is exact (checked over all 2^32 inputs on the host: tools/microbench/unit_float_check.c) --; 2 = RN(u) 2^-32, two
// instructions, which rounds 2^25 of the 2^32 inputs -- the ties with an even mantissa -- one ulp down)
#ifndef I3RC_UNIT_FLOAT
#define I3RC_UNIT_FLOAT 0
#endif
__host__ __device__ inline float u32_to_unit_float(uint32_t u) {
#if I3RC_UNIT_FLOAT == 1
  const double d = (double)u;
  return (float)__builtin_bit_cast(double, __builtin_bit_cast(uint64_t, d) | 1ull) * 2.3283064365386963e-10f;
#elif I3RC_UNIT_FLOAT == 2
  return (float)u * 2.3283064365386963e-10f;
#else
  return (float)((double)u * (1.0 / 4294967295.0));
#endif
}

// Per-lane stream.  begin_event() makes one Philox block (4 deviates) for the whole wavefront at ONE program point,
// so the 10-round block function runs once per event with every lane active.  (Refilling lazily ran the block
// function at ~9 % lane utilisation, because lanes drift to different buffer phases and every call site refills for
// whoever happens to be empty.)  The four words of an event's block have fixed roles:
//   first()   scattering angle  | start x of a new photon | cosine of a surface reflection
//   second()  azimuth           | start y                 | azimuth of a surface reflection
//   path()    optical depth to the next event
//   spare()   Russian roulette
// so that no run-time cursor has to be consulted for them.  Whatever else an event needs (component choice, the
// max-cross-section test, a retry) comes from next(), a cursor over further blocks of the same photon.  The
// local-estimate rays of an event draw from blocks of their own: same photon and block number, direction + 1 in the
// fourth counter word (kernels.hpp, ray mode).  Which word serves which purpose is this code's own convention: the production streams are not the
// reference's Mersenne Twister sequence anyway, and every production kernel follows the same convention (tests
// compare them photon by photon).  The replay stream hands out the reference's deviates in the reference's order.
// BATCHED (fused multi-batch launches, kernels.hpp): the lanes of a wave carry photons of DIFFERENT batches of a driver's
// loop -- the key's second word is seed1 + the lane's batch number (the driver's seed = (/iseed, batch/)), a vector value:
// its ten round keys cost ten vector adds per block, the price of filling one batch's tail with the next batch's photons.
template <bool BATCHED>
struct PhiloxStreamT {
  static constexpr bool kReplay = false;
  static constexpr bool kBatched = BATCHED;
  uint32_t k0, k1;           // key: the same for every photon of a launch (wave-uniform, lives in scalar registers)
  uint32_t batch;            // BATCHED: batch of this lane's photon, relative to the launch's first (else unused, 0)
  uint32_t id_lo, id_hi, block;
  uint32_t e0, e1, e2, e3;   // the event's block
  uint32_t b0, b1, b2, b3;   // block behind next()
  int have;                  // unused words of that block: next() hands out b[4 - have]
  uint32_t used;             // deviates consumed by this lane (all its photons; BATCHED: since the lane's last hand-over)

  // once per lane, in uniform control flow
  __device__ inline void init(uint32_t seed0, uint32_t seed1) {
    k0 = seed0; k1 = seed1; batch = 0u;
    id_lo = id_hi = 0u; block = 0u; have = 0; used = 0u; b0 = b1 = b2 = b3 = 0u; e0 = e1 = e2 = e3 = 0u;
  }
  // next photon of this lane
  __device__ inline void start(uint64_t photon) {
    id_lo = (uint32_t)photon; id_hi = (uint32_t)(photon >> 32);
    block = 0u; have = 0;
  }
  __device__ inline void start(uint64_t photon, uint32_t photonBatch) { start(photon); batch = photonBatch; }
  __device__ inline uint32_t take_used() { const uint32_t u = used; used = 0u; return u; }   // BATCHED: per-batch hand-over
  __device__ inline void close() {}
  // Philox coordinates of the event in progress (its block was made by begin_event): a local-estimate ray of this
  // event draws from the block with the same first three counter words and its direction number + 1 in the fourth
  __device__ inline uint32_t photon_lo() const { return id_lo; }
  __device__ inline uint32_t photon_hi() const { return id_hi; }
  __device__ inline uint32_t event_block() const { return block - 1u; }
  __device__ inline uint32_t lane_batch() const { return BATCHED ? batch : 0u; }   // fused launches: batch of this lane's photon
  __device__ inline uint32_t draws_of_photon() const { return 0u; }   // per-photon records are a replay-stream feature
  // deviates consumed by this lane so far (kernel epilogue)
  __device__ inline uint32_t total() const { return used; }
  __device__ inline Philox4 make_block() {
    // The key is wave-uniform; the empty asm makes it opaque here so that the ten round keys (k + r W) are made by
    // scalar adds next to their use instead of being hoisted out of the photon loop into twenty scalar registers
    // (which then spill).
    uint32_t s0 = k0, s1 = k1;
    asm volatile("" : "+s"(s0), "+s"(s1));
    const Philox4 o = philox4x32_10(id_lo, id_hi, block, 0u, s0, BATCHED ? s1 + batch : s1);
    block++;
    return o;
  }
  __device__ inline void begin_event() {
    const Philox4 o = make_block();
    e0 = o.v[0]; e1 = o.v[1]; e2 = o.v[2]; e3 = o.v[3];
  }
  __device__ inline void count_draws(uint32_t n) { used += n; }   // deviates drawn elsewhere on this lane's account (shadow rays)
  __device__ inline float first()  { used++; return u32_to_unit_float(e0); }
  __device__ inline float second() { used++; return u32_to_unit_float(e1); }
  __device__ inline float path()   { used++; return u32_to_unit_float(e2); }
  __device__ inline float spare()  { used++; return u32_to_unit_float(e3); }
  __device__ inline float next() {
    if (__builtin_expect(have == 0, 0)) {   // inline: an out-of-line call costs scratch spills at every site (measured -13 %)
      const Philox4 o = make_block();
      b0 = o.v[0]; b1 = o.v[1]; b2 = o.v[2]; b3 = o.v[3];
      have = 4;
    }
    const uint32_t u = have == 4 ? b0 : (have == 3 ? b1 : (have == 2 ? b2 : b3));
    have--;
    used++;
    return u32_to_unit_float(u);
  }
  // One deviate from a block of its own, without the cursor.  The specialised kernels draw outside an event's block only in
  // the all but impossible retry of a surface reflection's cosine (a deviate of exactly 0): using this there keeps the
  // cursor's five registers out of their photon loop.
  __device__ inline float fresh() { const Philox4 o = make_block(); used++; return u32_to_unit_float(o.v[0]); }
  // Azimuth for next_direct (:2099-2103).  The reference rejection-samples a point of the unit disc (2 deviates per
  // try, 21 % retries) only to get a uniformly distributed azimuth without trigonometry.  On the GPU one deviate and
  // the hardware sin/cos (arguments in revolutions) give the same distribution with no divergent retry loop; d
  // carries the actual squared radius so next_direct's normalisation stays exact.
  __device__ inline void disc_point(float &ax, float &ay, float &d) {
    const float turn = second();
    ax = __builtin_amdgcn_cosf(turn);
    ay = __builtin_amdgcn_sinf(turn);
    d = ax * ax + ay * ay;
  }
};

// (types of their own, not aliases: kernel names -- rocprofv3, tools/kernel_resources.py -- keep the plain form)
struct PhiloxStream : PhiloxStreamT<false> {};
struct PhiloxBatchStream : PhiloxStreamT<true> {};

// Test stream: deviates come from a buffer (the reference's MT19937 floats); see i3rc_hip_run_replay.
struct ReplayStream {
  static constexpr bool kReplay = true;   // consume deviates exactly where the reference does
  static constexpr bool kBatched = false;
  const float *buf;
  int64_t pos, end;
  int64_t photonStart;
  uint32_t closed;           // deviates consumed by the photons this lane has finished
  __device__ inline void init(const float *b, int64_t e) { buf = b; end = e; pos = photonStart = 0; closed = 0u; }
  __device__ inline void start(int64_t p) { pos = p; photonStart = p; }
  __device__ inline uint32_t draws_of_photon() const { return (uint32_t)(pos - photonStart); }
  __device__ inline void close() { closed += draws_of_photon(); photonStart = pos; }
  __device__ inline uint32_t total() const { return closed; }
  __device__ inline void begin_event() {}
  __device__ inline void count_draws(uint32_t) {}
  __device__ inline uint32_t photon_lo() const { return 0u; }   // (the replay build keeps the nested local estimate)
  __device__ inline uint32_t photon_hi() const { return 0u; }
  __device__ inline uint32_t event_block() const { return 0u; }
  __device__ inline uint32_t lane_batch() const { return 0u; }
  __device__ inline float next() {
    // a photon that parts from the reference's path (1-ulp differences of log / cos / ...) may ask for more deviates
    // than were recorded: past the end the recorded ones are used again from the start (a constant would be
    // degenerate: a rejection loop fed with 0.5 yields the zero vector and a NaN direction)
    const float r = buf[end > 0 ? pos % end : 0];
    pos++;
    return r;
  }
  __device__ inline float fresh() { return next(); }
  // the reference draws everything from one sequence: the roles of PhiloxStream are plain draws, in call order
  __device__ inline float first() { return next(); }
  __device__ inline float second() { return next(); }
  __device__ inline float path() { return next(); }
  __device__ inline float spare() { return next(); }
  // the reference's own rejection sampling, deviate for deviate (next_direct :2098-2103)
  __device__ inline void disc_point(float &ax, float &ay, float &d) {
    d = 2.0f;
    while (d > 1.0f) {
      ax = 1.0f - 2.0f * next();
      ay = 1.0f - 2.0f * next();
      d = ax * ax + ay * ay;
    }
  }
};

}  // namespace i3rc
