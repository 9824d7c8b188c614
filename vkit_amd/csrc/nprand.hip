// The numpy Generator streams of the noise operators, drawn on the device bit for bit.
//
// Reference call sites: vkit/mechanism/distortion/photometric/noise.py:44-54 (gaussion_noise:
// np.round(rng.normal(0, std, shape)).astype(int16)), :160-190 (speckle_noise: rng.normal(0, std, shape) in float64),
// :100-157 (impulse_noise: rng.choice((0, 1, 2), size, p) = one uniform double per pixel against a 3-entry cdf).
// The arithmetic lives in numpy (a dependency, not under /root/reference): PCG64 + the 256-layer ziggurat of
// numpy/random/src/distributions/distributions.c, restated in oracle/np_random.c and pinned there against numpy itself.
//
// What makes the stream serial on the host and how it is cut here
// ---------------------------------------------------------------
// Raw 64-bit draw k of a PCG64 generator is a pure function of (state, inc, k): the 128-bit LCG jumps ahead in
// O(log k), and a lane that owns every 64th draw advances with ONE 128-bit multiply-add by the constant (A^64, inc * G_64).
// The ziggurat consumes a VARIABLE number of draws per sample: 1 in 98.5 % of the attempts, 2 for a wedge test (the
// sample may then be rejected), 1 + 2m in the tail.  Which draws START an attempt is the serial part.  It is resolved
// speculatively: every draw position evaluates "the attempt that would start here" (length, accepted?, value); within a
// 64-draw round the true starts follow from a scalar walk over the few multi-draw attempts (ballot mask), and between
// rounds / tiles only one number travels: how many leading positions the previous attempt already consumed.
//
//   k_np_tile_states  one lane per tile: LCG state at the tile's first draw (binary jump with the constant powers of A)
//   k_np_scan         pass 1, one wavefront per tile of 64 * kRounds draws, assuming no carry-in: number of samples the
//                     tile emits, its carry-out, and the start / emit masks of its first round
//   k_np_resolve      one workgroup per stream: a tile's carry-in is its predecessor's carry-out; when that position is a
//                     start of the tile's speculative chain (all but ~2e-4 of the tiles) the chain merges there and count
//                     and carry-out follow from pass 1's masks; the rare other tiles are re-simulated in order by one
//                     wavefront; then an exclusive scan of the counts = the index of every tile's first sample
//   k_np_emit         pass 2, same walk with the true carry-in, samples written at their final index
//
// libm: the wedge test compares against exp(), the tail draws go through log1p().  The device versions are within an
// ulp or two of glibc's, not identical; every decision and every emitted integer that could depend on those last bits
// (|difference| inside a margin ~2^10 times the worst-case error) raises the job's AMBIGUOUS flag instead of guessing,
// and the caller redraws that plane with numpy on the host.  Expected rate: < 1e-6 per 2048^2 plane.
#include "vkx_internal.h"
#include "np_ziggurat.h"

#include <string.h>
#include <algorithm>

namespace {

typedef unsigned __int128 u128;
#define VKX_GLOBAL __attribute__((address_space(1)))

// Rounds of 64 draws per tile.  The attempts of a tile that are not a fast accept (1.5 % of its draws) are evaluated densely, one
// per lane, and their starts resolved by one pass over the lanes: with 16 rounds those passes ran at a quarter of the
// wavefront's width (15 events); 48 rounds give them 46 +- 7 events -- under the 64 the lane-parallel resolution takes -- and
// spread the per-tile fixed work (state jump, start resolution, bookkeeping) over three times the draws.
#ifndef VKX_NP_ROUNDS
#define VKX_NP_ROUNDS 48
#endif
constexpr int kRounds = VKX_NP_ROUNDS;    // even, <= 64 (lane r holds the masks of round r)
static_assert(kRounds % 2 == 0 && kRounds >= 2 && kRounds <= 64, "rounds per tile");
constexpr int kTile = 64 * kRounds;       // raw draws per tile
constexpr double kNorR = 3.6541528853610087963519472518;
constexpr double kNorInvR = 0.27366123732975827203338247596;

#define PCG_MULT ((((u128)0x2360ED051FC65DA4ull) << 64) | (u128)0x4385DF649FCCF645ull)

// Stream-independent jump constants: s_{k + j} = A^j s_k + inc * G_j with G_j = 1 + A + ... + A^(j-1) (mod 2^128).
struct JumpTabs {
    uint64_t pow2[64][4];    // j = 2^i: A^j lo, hi, G_j lo, hi
    uint64_t lane[64][4];    // j = l + 1
    uint64_t a64[2], g64[2];
    uint64_t a128[2], g128[2];
};
static JumpTabs g_jump_host;
static bool g_jump_host_ready = false;

static void jump_consts(u128 j, u128 *a, u128 *g)
{
    u128 acc_mult = 1, acc_plus = 0, cur_mult = PCG_MULT, cur_plus = 1;
    while (j > 0) {
        if (j & 1) {
            acc_mult *= cur_mult;
            acc_plus = acc_plus * cur_mult + cur_plus;
        }
        cur_plus = (cur_mult + 1) * cur_plus;
        cur_mult *= cur_mult;
        j >>= 1;
    }
    *a = acc_mult;
    *g = acc_plus;
}

static void build_jump_tabs()
{
    if (g_jump_host_ready) return;
    auto put = [](uint64_t *w, u128 a, u128 g) {
        w[0] = (uint64_t)a; w[1] = (uint64_t)(a >> 64); w[2] = (uint64_t)g; w[3] = (uint64_t)(g >> 64);
    };
    u128 a, g;
    for (int i = 0; i < 64; i++) { jump_consts((u128)1 << i, &a, &g); put(g_jump_host.pow2[i], a, g); }
    for (int l = 0; l < 64; l++) { jump_consts((u128)(l + 1), &a, &g); put(g_jump_host.lane[l], a, g); }
    jump_consts(64, &a, &g);
    g_jump_host.a64[0] = (uint64_t)a; g_jump_host.a64[1] = (uint64_t)(a >> 64);
    g_jump_host.g64[0] = (uint64_t)g; g_jump_host.g64[1] = (uint64_t)(g >> 64);
    jump_consts(128, &a, &g);
    g_jump_host.a128[0] = (uint64_t)a; g_jump_host.a128[1] = (uint64_t)(a >> 64);
    g_jump_host.g128[0] = (uint64_t)g; g_jump_host.g128[1] = (uint64_t)(g >> 64);
    g_jump_host_ready = true;
}

struct NpTabs {              // uploaded once per context
    JumpTabs jump;
    uint64_t ki[256], wi[256], fi[256];
};

// One stream of a batch as the kernels see it.
struct NpJob {
    uint64_t state[2], inc[2];
    uint64_t c64[2];         // inc * G_64: the addend of a 64-draw stride
    uint64_t c128[2];        // inc * G_128
    long long n;             // samples wanted
    long long tile_base;     // index of the job's first tile in the batch-wide tile arrays
    int n_tiles;
    int kind, cn;
    double loc, scale;
    double margin;           // relative width of the exp() ambiguity window (2^-42; VKX_NP_DEBUG_WIDE_MARGIN: 1)
    double cdf[3];
    const uint8_t *src;
    void *dst;
    int16_t *rec;            // compacted int16 records of the job's tiles, kSlot elements per tile (scratch, or the caller's tile buffer)
    uint2 *table;            // VKX_NP_NORMAL_TILES: [n_tiles + 1] x (index of the tile's first sample, its first valid slot element)
};

struct TileInfo {            // pass 1, assuming carry-in 0
    uint32_t count0, out0;   // out0: bit 31 = the tile holds a tail sample
    uint64_t start0, emit0;  // round 0: positions that start an attempt / that emit a sample
};
constexpr uint32_t kIrregular = 0x80000000u;   // TilePlan::c_in: the carry-in is not a start of the recorded chain
struct TilePlan {
    unsigned long long prefix;   // index of the tile's first sample
    uint32_t c_in, count;
};

__device__ __forceinline__ u128 mk128(const uint64_t *w) { return ((u128)w[1] << 64) | (u128)w[0]; }

__device__ __forceinline__ uint64_t pcg_out(u128 s)
{
    const uint64_t hi = (uint64_t)(s >> 64), lo = (uint64_t)s;
    const uint64_t x = hi ^ lo;
    const unsigned r = (unsigned)(hi >> 58);
    return (x >> r) | (x << ((64 - r) & 63));
}
// s -> a s + c (mod 2^128) for a wave-uniform (a, c): the stride step of the draw pass, 27 of its 52 instructions per round as the
// compiler expands the __int128 expression (4 v_mul_lo + 6 v_mad_u64_u32 + 2 v_add3 + 4 v_mov + 2 64-bit adds + 4 carry adds: gfx950
// wants 64-bit operands in even-aligned register pairs, so every "high half of a product, zero extended" costs a move).  Here by
// columns: the products of limb columns 0, 1 and 2 as three kinds of 64-bit accumulators that cannot overflow (a 32 x 32 product
// plus two 32-bit addends fits 64 bits) or may wrap (column 2: bits 128 and up are dropped anyway), the four products of column 3
// as a multiply-add chain of which the low word counts, and ONE carry chain over 32-bit halves, which need no alignment: 16.
struct LcgStride {
    uint32_t a0, a1, a2, a3;      // multiplier limbs
    uint64_t c0, c1, c23;         // addend: limb 0, limb 1 (zero extended), limbs 2-3
};
__device__ __forceinline__ LcgStride lcg_stride(u128 a, u128 c)
{
    LcgStride k;
    k.a0 = (uint32_t)a; k.a1 = (uint32_t)(a >> 32); k.a2 = (uint32_t)(a >> 64); k.a3 = (uint32_t)(a >> 96);
    k.c0 = (uint32_t)c; k.c1 = (uint32_t)(c >> 32); k.c23 = (uint64_t)(c >> 64);
    return k;
}
// (all operands in VGPRs: see the note on "s" operands in fused.hip; the carry-out the VOP3B encoding asks for goes nowhere)
__device__ __forceinline__ uint64_t mad_u64_u32_v(uint32_t a, uint32_t b, uint64_t c)
{
    uint64_t d, carry;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ uint64_t mul_u64_u32_v(uint32_t a, uint32_t b)
{
    uint64_t d, carry;
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(d), "=s"(carry) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ u128 lcg_step(u128 s, const LcgStride &k)
{
    const uint32_t s0 = (uint32_t)s, s1 = (uint32_t)(s >> 32), s2 = (uint32_t)(s >> 64), s3 = (uint32_t)(s >> 96);
    const uint64_t E = (uint64_t)k.a0 * s0 + k.c0;            // bits 0 .. 63
    const uint64_t O1 = (uint64_t)k.a0 * s1 + k.c1;           // bits 32 .. 95
    const uint64_t O2 = (uint64_t)k.a1 * s0;
    uint64_t F = (uint64_t)k.a0 * s2 + k.c23;                 // bits 64 .. 127
    F += (uint64_t)k.a1 * s1;
    F += (uint64_t)k.a2 * s0;
    uint64_t G = mul_u64_u32_v(k.a0, s3);                     // bits 96 .. 127: the low word
    G = mad_u64_u32_v(k.a1, s2, G);
    G = mad_u64_u32_v(k.a2, s1, G);
    G = mad_u64_u32_v(k.a3, s0, G);
    unsigned ca, cb, cc, cd;
    const uint32_t t = __builtin_addc((uint32_t)(E >> 32), (uint32_t)O1, 0u, &ca);
    const uint32_t r1 = __builtin_addc(t, (uint32_t)O2, 0u, &cb);
    const uint32_t u = __builtin_addc((uint32_t)F, (uint32_t)(O1 >> 32), ca, &cc);
    const uint32_t r2 = __builtin_addc(u, (uint32_t)(O2 >> 32), cb, &cd);
    unsigned ce;
    const uint32_t w = __builtin_addc((uint32_t)(F >> 32), (uint32_t)G, cc, &ce);
    const uint32_t r3 = __builtin_addc(w, 0u, cd, &ce);
    return ((u128)r3 << 96) | ((u128)r2 << 64) | ((u128)r1 << 32) | (uint32_t)E;
}

__device__ __forceinline__ double u2dbl(uint64_t u) { return (double)(long long)(u >> 11) * (1.0 / 9007199254740992.0); }

// What the samples become.  A draw is reduced to the emitter's `Val` as soon as it exists (an int16 step for the rounded
// kinds: 16 of them fit 8 registers while the tile waits for its true starts); pos = index of the sample in C order.
__device__ __forceinline__ int rounded_step(const NpJob &job, double z, bool inexact, uint32_t &flags)
{
    const double v = job.scale * z;            // 0 + std * z; np.round = rint, C cast to int16
    if (inexact) {
        const double f = v - floor(v);
        if (fabs(f - 0.5) < 1e-9) flags |= VKX_NP_AMBIGUOUS;
    }
    // |v| < 2^31: adding 1.5 * 2^52 leaves rint(v) (round half to even) in the low dword
    return (int16_t)(int)__double_as_longlong(v + 6755399441055744.0);
}
struct EmitNone {
    typedef int Val;
    typedef int16_t Store;
    __device__ static Val make(const NpJob &, double, bool, uint32_t &) { return 0; }
    __device__ static void store(const NpJob &, void *, long long, Val, bool, uint32_t &) {}
};
struct EmitI16 {   // np.round(0 + std * z).astype(int16)
    static constexpr bool kCheckAtStore = false;
    typedef int Val;
    typedef int16_t Store;
    __device__ static Val make(const NpJob &job, double z, bool inexact, uint32_t &flags) { return rounded_step(job, z, inexact, flags); }
    __device__ static void store(const NpJob &, void *dst, long long pos, Val k, bool, uint32_t &) { ((int16_t VKX_GLOBAL *)dst)[pos] = (int16_t)k; }
};
struct EmitAddU8 {   // clip(int16(px) + noise, 0, 255): the whole gaussion_noise operator
    static constexpr bool kCheckAtStore = false;
    typedef int Val;
    typedef int16_t Store;
    __device__ static Val make(const NpJob &job, double z, bool inexact, uint32_t &flags) { return rounded_step(job, z, inexact, flags); }
    __device__ static void store(const NpJob &job, void *dst, long long pos, Val k, bool, uint32_t &)
    {
        const int s = (int16_t)((int)((const uint8_t VKX_GLOBAL *)job.src)[pos] + k);
        ((uint8_t VKX_GLOBAL *)dst)[pos] = (uint8_t)vkd::clamp_u8(s);
    }
};
struct EmitSpeckle {   // uint8(clip(px + px * (0 + std * z), 0, 255)) in float64
    static constexpr bool kCheckAtStore = true;
    typedef double Val;
    typedef double Store;
    __device__ static Val make(const NpJob &, double z, bool, uint32_t &) { return z; }
    __device__ static void store(const NpJob &job, void *dst, long long pos, Val z, bool inexact, uint32_t &flags)
    {
        const double noise = 0.0 + job.scale * z;
        const double m = (double)((const uint8_t VKX_GLOBAL *)job.src)[pos];
        const double t = m * noise;
        double r = m + t;
        // a tail draw carries log1p's last bits: the pixel is in doubt only when px + px * noise sits on an integer boundary
        // (a zero pixel stays zero whatever the noise)
        if (inexact && m != 0.0 && fabs(r - rint(r)) < 1e-9) flags |= VKX_NP_AMBIGUOUS;
        r = r < 0.0 ? 0.0 : (r > 255.0 ? 255.0 : r);
        ((uint8_t VKX_GLOBAL *)dst)[pos] = (uint8_t)(int)r;
    }
};

// Per-wavefront LDS workspace of a tile walk.
constexpr int kEvCap = kRounds <= 48 ? 96 : 128;   // attempts per tile that are not a fast accept: 0.96 per round (48 rounds: 46 +- 7)
template <typename Val, bool WITH_VAL>
struct WaveWork {
    uint64_t ev_s[kEvCap][2];    // LCG state of the event's draw; after evaluation [0] holds the bits of a tail sample
    uint16_t ev_pos[kEvCap];     // 64 * round + lane
    uint16_t ev_info[kEvCap];    // draws consumed | emits << 8 | tail << 9
    uint64_t slow[kRounds];      // per round: the lanes whose attempt is not a fast accept
    uint64_t semit[kRounds];     // per round: the events that emit a sample, as a lane mask
    uint64_t cov[kRounds];       // per round: the draws consumed by an attempt that started earlier
    // kEmit / kCompact: what every draw would emit as a fast accept (int16 steps, or the float64 draw); kCompact squeezes the
    // tile's samples together in place and sends them to the tile's slot as 16-byte groups
    __attribute__((aligned(16))) Val val[WITH_VAL ? kRounds : 1][64];
};

__device__ __forceinline__ uint64_t rfl64(uint64_t v)
{
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ int mbcnt64(uint64_t m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// The walk of one wavefront over one tile.  `base` = LCG state before the tile's first draw, `c_in` = leading draws
// already consumed by the previous tile's last attempt.  Lane l owns draws 64 r + l of the tile.
//   phase 1  all rounds, branch-free but for the event push: draw, ziggurat layer lookup, value of a fast accept; the
//            few draws whose attempt needs more (wedge test, tail) are queued with their LCG state
//   phase 2  the queue is evaluated densely, one event per lane: exp() / log1p() run once per tile instead of once per
//            round with one active lane
//   phase 3  per round: the scalar walk over the multi-draw attempts that gives the true starts, then the emission
//   MODE kCount   the walk only: samples the tile yields and its carry-out for the given carry-in
//   MODE kRecord  carry-in 0: besides the count, what every draw would emit (`rec_val`, 1024 per tile, in draw order) and
//                 the 16 emit masks (`rec_mask`) go to global memory for k_np_place
//   MODE kEmit    the samples are stored at their final index of `emit_dst`
//   MODE kCompact carry-in 0: the samples the tile yields, squeezed together in LDS, go to the tile's slot `rec_val` as whole
//                 16-byte groups (int16 emitters; the consumer drops the leading samples a carry-in takes away)
enum { kCount = 0, kRecord = 1, kEmit = 2, kCompact = 3 };

__device__ __forceinline__ uint32_t lds_offset(const void *p)
{
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}

// The samples of a tile parked in draw order (`flat[64 r + l]` = what draw l of round r emits), lane r < kRounds holding round
// r's emit mask: squeezed together IN PLACE -- a sample's rank never exceeds its draw position, rounds are handled in order
// and a round reads its 64 elements before it writes.  Per round two v_readlane, two v_mbcnt whose addend carries the running
// count, one address op, a 2-byte LDS read and a write masked through exec.  Returns the number of samples.
__device__ __forceinline__ uint32_t compact_tile_lds(int16_t *flat, uint64_t m)
{
    const uint32_t lane = (uint32_t)__lane_id();
    const uint32_t base = lds_offset(flat);
    uint32_t run = 0;
    // eight rounds' reads are in flight before the first of their writes (a round's write lands at or below the round's own
    // elements, never on a later round's): one LDS round trip per eight rounds instead of one per round
    constexpr int kBatch = 8;
    static_assert(kRounds % kBatch == 0 || kRounds < kBatch, "rounds per compaction batch");
#pragma unroll
    for (int r0 = 0; r0 < kRounds; r0 += kBatch) {
        uint32_t v[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; j++)
            if (r0 + j < kRounds) v[j] = (uint16_t)flat[64 * (r0 + j) + lane];
#pragma unroll
        for (int j = 0; j < kBatch; j++) {
            const int r = r0 + j;
            if (r >= kRounds) break;
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)m, r), hi = (uint32_t)__builtin_amdgcn_readlane((int)(m >> 32), r);
            const uint32_t rank = (uint32_t)__builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, run));
            const uint32_t addr = base + 2 * rank;
            const uint64_t mask = ((uint64_t)hi << 32) | lo;
            uint64_t saved;
            asm volatile("s_mov_b64 %[sv], exec\n\ts_mov_b64 exec, %[m]\n\tds_write_b16 %[a], %[v]\n\ts_mov_b64 exec, %[sv]"
                         : [sv] "=&s"(saved) : [m] "s"(mask), [a] "v"(addr), [v] "v"(v[j]) : "memory");
            run += (uint32_t)__builtin_popcountll(mask);
        }
    }
    __builtin_amdgcn_wave_barrier();
    return run;
}

template <class Emit, int MODE>
__device__ void walk_tile(const NpJob &job, const JumpTabs &g_jump, const uint4 *__restrict__ zig /* LDS: ki | wi */, const double *__restrict__ fi /* LDS */,
                          WaveWork<typename Emit::Store, MODE == kEmit || MODE == kCompact> &ws, u128 base, uint32_t c_in, long long prefix,
                          long long draw_base, typename Emit::Store VKX_GLOBAL *rec_val, uint64_t VKX_GLOBAL *rec_mask, void *emit_dst,
                          uint32_t &count_out, uint32_t &carry_out, uint64_t &start0, uint64_t &emit0, bool &has_tail, uint32_t &flags,
                          unsigned long long *draws_used)
{
    const int lane = __lane_id();
    const u128 inc = mk128(job.inc);
    const u128 a64 = mk128(g_jump.a64), c64 = mk128(job.c64);
    // wave-uniform by construction; said explicitly so that the start walk below stays on the scalar unit
    c_in = (uint32_t)__builtin_amdgcn_readfirstlane((int)c_in);
    prefix = (long long)rfl64((uint64_t)prefix);
    draw_base = (long long)rfl64((uint64_t)draw_base);
    // state after draw `lane` of the tile has been stepped
    u128 s = mk128(&g_jump.lane[lane][0]) * base + mk128(&g_jump.lane[lane][2]) * inc;
    uint32_t nev = 0;
    if (lane < kRounds) ws.semit[lane] = 0;
    // two rounds per iteration from two independent generator states (draws 64 r + l and 64 (r + 1) + l, both striding
    // 128): twice the instruction-level parallelism for the long dependent chain state -> draw -> table -> compare
    auto round = [&](int r, const u128 &st) {
        const uint64_t u = pcg_out(st);
        const int idx = (int)(u & 0xff);
        const uint64_t rabs = (u >> 9) & 0x000fffffffffffffull;
        const uint4 e = zig[idx];
        const uint64_t ki = ((uint64_t)e.y << 32) | e.x;
        if (MODE != kCount) {
            const double wi = __longlong_as_double(((long long)e.w << 32) | e.z);
            // rabs < 2^52: exact conversion through the exponent trick
            double x = (__longlong_as_double((long long)(0x4330000000000000ull | rabs)) - 4503599627370496.0) * wi;
            // x >= 0: the sign (bit 8 of the draw) is or-ed into the float64 sign bit
            x = __longlong_as_double(__double_as_longlong(x) | ((long long)((uint32_t)u << 23 & 0x80000000u) << 32));
            const typename Emit::Store v = (typename Emit::Store)Emit::make(job, x, false, flags);
            if (MODE == kRecord) rec_val[64 * r + lane] = v;
            else if (MODE == kEmit || MODE == kCompact) ws.val[r][lane] = v;
        }
        const uint64_t slow = __ballot(rabs >= ki);
        if (lane == 0) ws.slow[r] = slow;
        if (slow) {
            if (rabs >= ki) {
                const uint32_t slot = nev + (uint32_t)mbcnt64(slow);
                if (slot < (uint32_t)kEvCap) {
                    ws.ev_s[slot][0] = (uint64_t)st;
                    ws.ev_s[slot][1] = (uint64_t)(st >> 64);
                    ws.ev_pos[slot] = (uint16_t)(64 * r + lane);
                }
            }
            nev += (uint32_t)__builtin_popcountll(slow);
        }
    };
    const LcgStride stride128 = lcg_stride(mk128(g_jump.a128), mk128(job.c128));
    u128 s2 = a64 * s + c64;
#pragma unroll 1
    for (int r = 0; r < kRounds; r += 2) {
        round(r, s);
        round(r + 1, s2);
        s = lcg_step(s, stride128);
        s2 = lcg_step(s2, stride128);
    }
    if (nev > (uint32_t)kEvCap) {   // never observed; the job is redrawn on the host
        flags |= VKX_NP_SHORT;
        nev = kEvCap;
    }
    __builtin_amdgcn_wave_barrier();
    if (MODE == kRecord) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tail samples below overwrite phase 1's stores
    has_tail = false;
    // phase 2
#pragma unroll
    for (int q = 0; q < (kEvCap + 63) / 64; q++) {
        if ((uint32_t)(64 * q) >= nev) break;
        const uint32_t ev = 64 * q + lane;
        if (ev < nev) {
            const u128 se = ((u128)ws.ev_s[ev][1] << 64) | ws.ev_s[ev][0];
            const int pos = ws.ev_pos[ev];
            const uint64_t u = pcg_out(se);
            const int idx = (int)(u & 0xff);
            const uint64_t rabs = (u >> 9) & 0x000fffffffffffffull;
            uint32_t info;
            if (idx != 0) {
                double x = (__longlong_as_double((long long)(0x4330000000000000ull | rabs)) - 4503599627370496.0) *
                           __longlong_as_double(((long long)zig[idx].w << 32) | zig[idx].z);
                if (u & 0x100) x = -x;
                const double un = u2dbl(pcg_out(PCG_MULT * se + inc));
                const double f1 = fi[idx], f0 = fi[idx - 1];
                const double lhs = (f0 - f1) * un + f1;
                const double t = -0.5 * x;
                const double y = t * x;                       // in [-6.7, 0]
                // most tests are decided by a float32 estimate of exp(y) (hardware exp2: relative error < 2^-20 with the
                // argument reduction lost in y * log2(e)); only those within 2^-14 of the threshold pay for the float64 exp
                const float r32 = __builtin_amdgcn_exp2f((float)y * 1.44269504f);
                const float l32 = (float)lhs;
                bool accept = l32 < r32;
                if (fabsf(l32 - r32) <= r32 * 0x1p-14f || job.margin >= 1.0) {
                    const double rhs = exp(y);
                    if (fabs(rhs - lhs) <= lhs * job.margin) flags |= VKX_NP_AMBIGUOUS;
                    accept = lhs < rhs;
                }
                info = 2u | (accept ? 0x100u : 0u);
            } else {
                u128 t = se;
                uint32_t len = 1;
                double xx;
                for (;;) {
                    t = PCG_MULT * t + inc;
                    const double u1 = u2dbl(pcg_out(t));
                    t = PCG_MULT * t + inc;
                    const double u2 = u2dbl(pcg_out(t));
                    len += 2;
                    xx = -kNorInvR * log1p(-u1);
                    const double yy = -log1p(-u2);
                    const double l2 = yy + yy, r2 = xx * xx;
                    if (fabs(l2 - r2) <= r2 * 0x1p-40) flags |= VKX_NP_AMBIGUOUS;
                    if (l2 > r2 || len > 200) break;
                }
                if (len > 200) flags |= VKX_NP_SHORT;
                const double z = ((rabs >> 8) & 1) ? -(kNorR + xx) : kNorR + xx;
                if (MODE == kRecord) {
                    rec_val[pos] = (typename Emit::Store)Emit::make(job, z, true, flags);
                } else if (MODE == kCompact) {
                    (&ws.val[0][0])[pos] = (typename Emit::Store)Emit::make(job, z, true, flags);
                } else if (MODE == kEmit) {
                    const typename Emit::Val v = Emit::make(job, z, true, flags);
                    uint64_t bits = 0;
                    memcpy(&bits, &v, sizeof v);
                    ws.ev_s[ev][0] = bits;
                }
                has_tail = true;
                info = len | 0x300u;
            }
            ws.ev_info[ev] = (uint16_t)info;
            if (info & 0x100u) atomicOr((unsigned long long *)&ws.semit[pos >> 6], 1ull << (pos & 63));
        }
    }
    __builtin_amdgcn_wave_barrier();
    has_tail = __ballot(has_tail) != 0;
    // phase 3: which attempts really start.  Every event (an attempt that is not a fast accept) consumes at least one
    // draw after its own; an event is void when an earlier valid one has consumed its draw.  The events are sorted by
    // position, so a void event always follows the valid event that covers it directly or through other void events:
    // when no event starts inside its predecessor's span (4 tiles out of 5) all are valid and nothing is serial.
    uint32_t P = 0xffffu, end = 0, info = 0;
    if ((uint32_t)lane < nev) {
        P = ws.ev_pos[lane];
        info = ws.ev_info[lane];
        end = P + (info & 0xffu);
    }
    uint64_t valid;
    uint32_t carry = c_in > (uint32_t)kTile ? c_in - kTile : 0u;
    if (nev <= 64) {
        const uint32_t prev_end = (uint32_t)__builtin_amdgcn_update_dpp((int)c_in, (int)end, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
        const uint64_t live = nev == 64 ? ~0ull : ((1ull << nev) - 1);
        const uint64_t touch = __ballot((uint32_t)lane < nev && P < prev_end);
        valid = live;
        if (touch) {
            uint32_t cover_end = c_in;
            valid = 0;
            for (uint32_t e = 0; e < nev; e++) {
                const uint32_t pe = (uint32_t)__builtin_amdgcn_readlane((int)P, (int)e);
                if (pe >= cover_end) {
                    valid |= 1ull << e;
                    cover_end = (uint32_t)__builtin_amdgcn_readlane((int)end, (int)e);
                }
            }
        }
        if (valid) {
            const uint32_t last_end = (uint32_t)__builtin_amdgcn_readlane((int)end, 63 - __builtin_clzll(valid));
            if (last_end > (uint32_t)kTile && last_end - kTile > carry) carry = last_end - kTile;
        }
        // the draws each valid event consumes after its own: positions P + 1 .. end - 1
        if (lane < kRounds) ws.cov[lane] = 0;
        __builtin_amdgcn_wave_barrier();
        if ((valid >> lane) & 1) {
            uint32_t p0 = P + 1, p1 = end < (uint32_t)kTile ? end : (uint32_t)kTile;     // [p0, p1)
            while (p0 < p1) {
                const uint32_t r = p0 >> 6, stop = (r + 1) << 6 < p1 ? (r + 1) << 6 : p1;
                const uint32_t lo = p0 & 63, n = stop - p0;
                const uint64_t bits = (n == 64 ? ~0ull : ((1ull << n) - 1)) << lo;
                atomicOr((unsigned long long *)&ws.cov[r], bits);
                p0 = stop;
            }
        }
        __builtin_amdgcn_wave_barrier();
    } else {
        // more than 64 events in one tile (12 sigma): the plain serial walk
        valid = 0;
        if (lane < kRounds) ws.cov[lane] = 0;
        __builtin_amdgcn_wave_barrier();
        uint32_t cover_end = c_in;
        for (uint32_t e = 0; e < nev; e++) {
            const uint32_t pe = ws.ev_pos[e], ee = pe + (ws.ev_info[e] & 0xffu);
            if (pe < cover_end) continue;
            cover_end = ee;
            if (lane == 0)
                for (uint32_t q = pe + 1; q < ee && q < (uint32_t)kTile; q++) ws.cov[q >> 6] |= 1ull << (q & 63);
        }
        if (cover_end > (uint32_t)kTile && cover_end - kTile > carry) carry = cover_end - kTile;
        __builtin_amdgcn_wave_barrier();
    }
    // lane r: the masks of round r
    uint64_t my_mask = 0, my_starts = 0;
    if (lane < kRounds) {
        uint64_t cov = ws.cov[lane];
        const uint32_t lo = 64u * lane;
        if (c_in > lo) cov |= c_in - lo >= 64 ? ~0ull : ((1ull << (c_in - lo)) - 1);
        const uint64_t slow = ws.slow[lane];
        my_starts = ~cov;
        my_mask = my_starts & (~slow | ws.semit[lane]);
    }
    start0 = rfl64(my_starts);
    emit0 = rfl64(my_mask);
    uint32_t count = (uint32_t)__builtin_popcountll(my_mask);
    count += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)count, 0x118 /* row_shr:8 */, 0xf, 0xf, true);
    count += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)count, 0x114 /* row_shr:4 */, 0xf, 0xf, true);
    count += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)count, 0x112 /* row_shr:2 */, 0xf, 0xf, true);
    count += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)count, 0x111 /* row_shr:1 */, 0xf, 0xf, true);
    {   // lane 15 of every row of 16 lanes holds the row's sum
        uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)count, 15);
        if (kRounds > 16) total += (uint32_t)__builtin_amdgcn_readlane((int)count, 31);
        if (kRounds > 32) total += (uint32_t)__builtin_amdgcn_readlane((int)count, 47);
        if (kRounds > 48) total += (uint32_t)__builtin_amdgcn_readlane((int)count, 63);
        count = total;
    }
    if (MODE == kEmit) {
        uint32_t done = 0, ebase = 0;
#pragma unroll 1
        for (int r = 0; r < kRounds; r++) {
            if (prefix + done >= job.n) break;   // everything wanted has been written
            const uint64_t emits = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(my_mask >> 32), r) << 32) |
                                   (uint32_t)__builtin_amdgcn_readlane((int)my_mask, r);
            const uint64_t slow = rfl64(ws.slow[r]);
            const bool mine = (emits >> lane) & 1;
            const long long pos = prefix + done + mbcnt64(emits);
            if (mine && pos < job.n) {
                typename Emit::Val z = (typename Emit::Val)ws.val[r][lane];
                bool inexact = false;
                int len = 1;
                if ((slow >> lane) & 1) {
                    const uint32_t ev = ebase + (uint32_t)mbcnt64(slow);
                    const uint32_t einfo = ws.ev_info[ev];
                    len = (int)(einfo & 0xff);
                    if (einfo & 0x200u) {
                        const uint64_t bits = ws.ev_s[ev][0];
                        memcpy(&z, &bits, sizeof z);
                        inexact = true;
                    }
                }
                Emit::store(job, emit_dst, pos, z, inexact, flags);
                if (pos == job.n - 1)      // write-through: read by the last workgroup of the kernel (np_results_out)
                    __hip_atomic_store(draws_used, (unsigned long long)(draw_base + 64 * r + lane + len), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            done += (uint32_t)__builtin_popcountll(emits);
            ebase += (uint32_t)__builtin_popcountll(slow);
        }
    }
    if (MODE == kRecord && lane < kRounds) rec_mask[lane] = my_mask;
    if constexpr (MODE == kCompact) {
        static_assert(sizeof(typename Emit::Store) == 2, "compact records are int16");
        int16_t *flat = (int16_t *)&ws.val[0][0];
        const uint32_t total = compact_tile_lds(flat, my_mask);        // == count
        const uint32_t groups = (total + 7) >> 3;                      // the last group's surplus elements are never read
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 *f16 = (const u32x4 *)flat;
        u32x4 VKX_GLOBAL *s16 = (u32x4 VKX_GLOBAL *)rec_val;
#pragma unroll
        for (int i = 0; i < kTile / 512; i++) {
            const uint32_t idx = (uint32_t)lane + 64u * i;
            if (idx < groups) s16[idx] = f16[idx];
        }
        __builtin_amdgcn_wave_barrier();
    }
    count_out = count;
    carry_out = carry;
}

__device__ __forceinline__ void load_tables(uint4 *zig, double *fi, const NpTabs *__restrict__ tabs)
{
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
        const uint64_t k = tabs->ki[i], w = tabs->wi[i];
        zig[i] = make_uint4((uint32_t)k, (uint32_t)(k >> 32), (uint32_t)w, (uint32_t)(w >> 32));
        fi[i] = __longlong_as_double((long long)tabs->fi[i]);
    }
    __syncthreads();
}

__device__ __forceinline__ int job_of_tile(const NpJob *jobs, int n_jobs, long long tile)
{
    int lo = 0, hi = n_jobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].tile_base <= tile) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// Small calls (one page of a host pipeline) carry their jobs in the kernel arguments: on a stream that shares the device
// with the multi-megabyte transfers of other pipeline lanes every extra dispatch (a descriptor copy, a memset) costs the
// pipeline tens of microseconds (tools/probes/pipe_np_trace.py).  The first workgroup leaves them in device memory for
// the kernels that follow, zeroes the results and the completion counter.
constexpr int kInlineJobs = 8;
struct NpJobPack { NpJob jobs[kInlineJobs]; };

template <bool INLINE>
__global__ void __launch_bounds__(256) k_np_tile_states(const NpJob *__restrict__ jobs_dev, NpJobPack pack, int n_jobs, long long total_tiles,
                                                        uint64_t *__restrict__ states /* [total_tiles][2] */, NpJob *__restrict__ jobs_out,
                                                        vkx_np_result *__restrict__ results, unsigned *__restrict__ done,
                                                        const NpTabs *__restrict__ tabs)
{
    const NpJob *jobs = INLINE ? pack.jobs : jobs_dev;
    if (blockIdx.x == 0) {
        if (INLINE) {
            const uint32_t *src = (const uint32_t *)pack.jobs;
            uint32_t *dst = (uint32_t *)jobs_out;
            for (unsigned i = threadIdx.x; i < (unsigned)n_jobs * (sizeof(NpJob) / 4); i += 256) dst[i] = src[i];
        }
        uint32_t *r = (uint32_t *)results;
        for (unsigned i = threadIdx.x; i < (unsigned)n_jobs * (sizeof(vkx_np_result) / 4); i += 256) r[i] = 0;
        if (threadIdx.x == 0) *done = 0;
    }
    const JumpTabs &g_jump = tabs->jump;
    const long long tile = (long long)blockIdx.x * 256 + threadIdx.x;
    if (tile >= total_tiles) return;
    const NpJob &job = jobs[job_of_tile(jobs, n_jobs, tile)];
    unsigned long long d = (unsigned long long)(tile - job.tile_base) * kTile;
    u128 s = mk128(job.state);
    const u128 inc = mk128(job.inc);
    for (int i = 0; d; i++, d >>= 1)
        if (d & 1) s = mk128(&g_jump.pow2[i][0]) * s + mk128(&g_jump.pow2[i][2]) * inc;
    states[2 * tile] = (uint64_t)s;
    states[2 * tile + 1] = (uint64_t)(s >> 64);
}

// The last workgroup of a call's final kernel hands the results to the host through the mapping of its page-locked
// buffer (results_host == nullptr: the caller's buffer is pageable, the host queues a copy instead).
__device__ __forceinline__ void np_results_out(unsigned *done, const vkx_np_result *results, vkx_np_result *results_host, int n_jobs)
{
    if (!results_host) return;
    __shared__ bool s_last;
    // this kernel's writes to `results` are device-scope atomics (flags) and write-through stores (draws): once a wavefront's
    // own stores are acknowledged they are visible to the sc1 loads below -- no L2 write-back fence (~2 us per workgroup)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(done, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!s_last) return;
    const uint32_t *src = (const uint32_t *)results;
    uint32_t *dst = (uint32_t *)results_host;
    for (unsigned i = threadIdx.x; i < (unsigned)n_jobs * (sizeof(vkx_np_result) / 4); i += blockDim.x)
        dst[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <class Emit>
__global__ void __launch_bounds__(256) k_np_draw(const NpJob *__restrict__ jobs, int n_jobs, long long total_tiles,
                                                 const uint64_t *__restrict__ states, TileInfo *__restrict__ info,
                                                 typename Emit::Store *__restrict__ rec_val, uint64_t *__restrict__ rec_mask,
                                                 vkx_np_result *__restrict__ results, const NpTabs *__restrict__ tabs)
{
    __shared__ uint4 zig[256];
    __shared__ double fi[256];
    __shared__ WaveWork<typename Emit::Store, false> work[4];
    load_tables(zig, fi, tabs);
    const JumpTabs &g_jump = tabs->jump;
    const long long n_waves = (long long)gridDim.x * 4;
    for (long long tile = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); tile < total_tiles; tile += n_waves) {
        const int j = job_of_tile(jobs, n_jobs, tile);
        const NpJob &job = jobs[j];
        uint32_t count, carry, flags = 0;
        uint64_t start0, emit0;
        bool has_tail;
        walk_tile<Emit, kRecord>(job, g_jump, zig, fi, work[threadIdx.x >> 6], mk128(&states[2 * tile]), 0u, 0, 0,
                                 (typename Emit::Store VKX_GLOBAL *)(rec_val + tile * kTile), (uint64_t VKX_GLOBAL *)(rec_mask + tile * kRounds),
                                 nullptr, count, carry, start0, emit0, has_tail, flags, nullptr);
        if (__lane_id() == 0) {
            TileInfo t;
            t.count0 = count; t.out0 = carry | (has_tail ? 0x80000000u : 0u); t.start0 = start0; t.emit0 = emit0;
            info[tile] = t;
        }
        if (flags) atomicOr(&results[j].flags, flags);
    }
}

// The draw pass of the int16 kinds: the same walk, but a tile's samples leave squeezed together (kCompact) for the tile's slot of
// kSlot elements -- the placement pass that used to read 2 bytes per RAW DRAW plus the emit masks, rank the samples and
// scatter them is gone: what follows only copies (k_np_apply), or nothing follows at all (VKX_NP_NORMAL_TILES: the chain
// kernel reads the slots).  kDrawWaves wavefronts share the ziggurat tables: 8 x 9.0 KB + 6 KB per workgroup, two workgroups
// (16 wavefronts at 112 VGPRs) per CU.
constexpr int kSlot = kTile + 16;        // elements per slot: the tile's samples, two lent by the next tile, 16-byte group rounding
#ifndef VKX_NP_DRAW_WAVES
#define VKX_NP_DRAW_WAVES 8
#endif
constexpr int kDrawWaves = VKX_NP_DRAW_WAVES;

template <class Emit>
__global__ void __launch_bounds__(64 * kDrawWaves) k_np_draw_compact(const NpJob *__restrict__ jobs, int n_jobs, long long total_tiles,
                                                                    const uint64_t *__restrict__ states, TileInfo *__restrict__ info,
                                                                    vkx_np_result *__restrict__ results, const NpTabs *__restrict__ tabs)
{
    __shared__ uint4 zig[256];
    __shared__ double fi[256];
    __shared__ WaveWork<int16_t, true> work[kDrawWaves];
    load_tables(zig, fi, tabs);
    const JumpTabs &g_jump = tabs->jump;
    const long long n_waves = (long long)gridDim.x * kDrawWaves;
    for (long long tile = (long long)blockIdx.x * kDrawWaves + (threadIdx.x >> 6); tile < total_tiles; tile += n_waves) {
        const int j = job_of_tile(jobs, n_jobs, tile);
        const NpJob &job = jobs[j];
        uint32_t count, carry, flags = 0;
        uint64_t start0, emit0;
        bool has_tail;
        walk_tile<Emit, kCompact>(job, g_jump, zig, fi, work[threadIdx.x >> 6], mk128(&states[2 * tile]), 0u, 0, 0,
                                  (int16_t VKX_GLOBAL *)(job.rec + (tile - job.tile_base) * kSlot), nullptr, nullptr,
                                  count, carry, start0, emit0, has_tail, flags, nullptr);
        if (__lane_id() == 0) {
            TileInfo t;
            t.count0 = count; t.out0 = carry | (has_tail ? 0x80000000u : 0u); t.start0 = start0; t.emit0 = emit0;
            info[tile] = t;
        }
        if (flags) atomicOr(&results[j].flags, flags);
    }
}

// One workgroup of 1024 per stream.
__global__ void __launch_bounds__(1024) k_np_resolve(const NpJob *__restrict__ jobs, const uint64_t *__restrict__ states,
                                                     const TileInfo *__restrict__ info_all, TilePlan *__restrict__ plan_all,
                                                     vkx_np_result *__restrict__ results, const NpTabs *__restrict__ tabs)
{
    __shared__ uint4 zig[256];
    __shared__ double fi[256];
    __shared__ WaveWork<int16_t, false> work[1];
    __shared__ unsigned long long part[1024];
    __shared__ uint32_t n_irregular;
    extern __shared__ uint64_t irregular[];     // one bit per tile
    const NpJob &job = jobs[blockIdx.x];
    const int T = job.n_tiles;
    const TileInfo *info = info_all + job.tile_base;
    TilePlan *plan = plan_all + job.tile_base;
    const uint64_t *st = states + 2 * job.tile_base;
    const int words = (T + 63) >> 6;
    if (threadIdx.x == 0) n_irregular = 0;
    load_tables(zig, fi, tabs);
    const JumpTabs &g_jump = tabs->jump;
    // phase A: every tile takes its predecessor's speculative carry-out
    for (int w = threadIdx.x >> 6; w < words; w += 16) {
        const int j = w * 64 + (threadIdx.x & 63);
        bool bad = false;
        if (j < T) {
            const uint32_t c = j ? (info[j - 1].out0 & 0x7fffffffu) : 0u;
            const TileInfo t = info[j];
            const bool ok = c < 64 && ((t.start0 >> c) & 1);
            TilePlan p;
            p.prefix = 0;
            p.c_in = c;
            p.count = ok ? t.count0 - (uint32_t)__builtin_popcountll(t.emit0 & ((1ull << c) - 1)) : 0u;
            plan[j] = p;
            bad = !ok;
        }
        const uint64_t m = __ballot(bad);
        if ((threadIdx.x & 63) == 0) {
            irregular[w] = m;
            if (m) atomicAdd(&n_irregular, (uint32_t)__builtin_popcountll(m));
        }
    }
    __syncthreads();
    // phase B: the tiles whose carry-in is not a start of their speculative chain, in order, by wavefront 0
    if (n_irregular && threadIdx.x < 64) {
        uint32_t flags = 0;
        for (int w = 0; w < words; w++) {
            for (;;) {
                __threadfence_block();
                const uint64_t m = ((volatile uint64_t *)irregular)[w];
                if (!m) break;
                const int b = __builtin_ctzll(m);
                const int j = w * 64 + b;
                const TileInfo t = info[j];
                const uint32_t c = ((volatile TilePlan *)plan)[j].c_in & ~kIrregular;
                const uint32_t out0 = t.out0 & 0x7fffffffu;
                uint32_t count, out;
                bool simulated = false;
                if (c < 64 && ((t.start0 >> c) & 1)) {
                    count = t.count0 - (uint32_t)__builtin_popcountll(t.emit0 & ((1ull << c) - 1));
                    out = out0;
                } else {
                    uint64_t s0, e0;
                    bool ht;
                    walk_tile<EmitNone, kCount>(job, g_jump, zig, fi, work[0], mk128(&st[2 * j]), c, 0, 0, nullptr, nullptr, nullptr, count, out,
                                                s0, e0, ht, flags, nullptr);
                    simulated = true;
                }
                if (threadIdx.x == 0) {
                    plan[j].count = count;
                    plan[j].c_in = c | (simulated ? kIrregular : 0u);
                    irregular[w] = m & ~(1ull << b);
                    if (j + 1 < T && out != out0) {
                        const TileInfo tn = info[j + 1];
                        const bool ok = out < 64 && ((tn.start0 >> out) & 1);
                        plan[j + 1].c_in = out;
                        plan[j + 1].count = ok ? tn.count0 - (uint32_t)__builtin_popcountll(tn.emit0 & ((1ull << out) - 1)) : 0u;
                        const int wn = (j + 1) >> 6;
                        const uint64_t bit = 1ull << ((j + 1) & 63);
                        irregular[wn] = ok ? (irregular[wn] & ~bit) : (irregular[wn] | bit);
                    }
                }
            }
        }
        if (flags && threadIdx.x == 0) atomicOr(&results[blockIdx.x].flags, flags);
    }
    __threadfence_block();
    __syncthreads();
    // phase C: exclusive scan of the counts
    const int per = (T + 1023) / 1024;
    const int j0 = threadIdx.x * per, j1 = min(T, j0 + per);
    unsigned long long sum = 0;
    for (int j = j0; j < j1; j++) sum += plan[j].count;
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const unsigned long long v = threadIdx.x >= d ? part[threadIdx.x - d] : 0ull;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    unsigned long long run = part[threadIdx.x] - sum;
    bool short_tile = false;
    for (int j = j0; j < j1; j++) {
        plan[j].prefix = run;
        // consumers of tile buffers assume that a row segment of 64 pixels (192 samples) meets at most two tiles and that a tile
        // can lend its predecessor two samples: a tile inside the wanted range that yields fewer (of 3 072 draws: never observed,
        // ~1e-3000) hands the stream to the host like any other refusal
        short_tile = short_tile || ((long long)run < job.n && j + 1 < T && plan[j].count < 192u);
        run += plan[j].count;
    }
    if (short_tile && job.kind == VKX_NP_NORMAL_TILES) atomicOr(&results[blockIdx.x].flags, VKX_NP_SHORT);
    if (threadIdx.x == 1023) {
        results[blockIdx.x].samples = part[1023];
        if ((long long)part[1023] < job.n) atomicOr(&results[blockIdx.x].flags, VKX_NP_SHORT);
    }
}

// The fast path of pass 2 for one tile: m = (lane r < 16) the emit mask of round r.
// Element-granular version (speckle: float64 records, byte traffic; a rare single-image operator).
template <class Emit>
struct Placer {
    struct Draws {};
    __device__ static Draws load(const typename Emit::Store *) { return Draws(); }
    __device__ static void place(const NpJob &job, const Draws &, const typename Emit::Store *__restrict__ rec, uint64_t m, long long prefix,
                                 typename Emit::Store *, uint32_t &flags)
{
    const typename Emit::Store VKX_GLOBAL *vals = (const typename Emit::Store VKX_GLOBAL *)rec;
    const int lane = __lane_id();
    uint32_t count = 0;
#pragma unroll 4
    for (int r = 0; r < kRounds; r++) {
        const uint64_t emits = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(m >> 32), r) << 32) |
                               (uint32_t)__builtin_amdgcn_readlane((int)m, r);
        const long long pos = prefix + count + mbcnt64(emits);
        if (((emits >> lane) & 1) && pos < job.n) Emit::store(job, job.dst, pos, (typename Emit::Val)vals[64 * r + lane], false, flags);
        count += (uint32_t)__builtin_popcountll(emits);
    }
}
};

// int16 records: sub-dword global accesses cost the texture path one lane per cycle, so the records are read as dwords
// (two consecutive draws per lane), compacted into LDS at their rank, and leave as whole dwords.
// `shift` = (index of the tile's first sample) mod (elements per dword): LDS slot i holds element (prefix - shift) + i.
struct TileDraws { uint32_t v[kRounds / 2]; };     // lane l: draws 128 q + 2 l, + 1 of the tile as int16 pairs
__device__ __forceinline__ TileDraws load_tile_i16(const int16_t *__restrict__ rec)
{
    const uint32_t VKX_GLOBAL *vals = (const uint32_t VKX_GLOBAL *)rec;
    const int lane = __lane_id();
    TileDraws d;
#pragma unroll
    for (int q = 0; q < kRounds / 2; q++) d.v[q] = vals[64 * q + lane];
    return d;
}
// The compaction, with the records parked in the staging row first (8 elements further up, so that a rank -- at most
// shift + position, shift < 8 -- never overtakes a record that has not been read): per round two v_readlane (the round's mask
// lives in lane r), two v_mbcnt whose addend carries the running count, one address op, a 2-byte LDS read and a write that is
// masked through exec instead of a per-lane bit test -- a quarter of the VALU instructions of a compaction out of registers
// (rank of each of a lane's two draws by popcounts and selects), in a pass that shares the SIMDs with the draw pass of the
// next chunk.
constexpr int kRawOff = 8;
__device__ __forceinline__ uint32_t stage_tile_lds(const TileDraws &d, uint64_t m, uint32_t shift, int16_t *st)
{
    const uint32_t lane = (uint32_t)__lane_id();
    uint32_t *raw32 = (uint32_t *)(st + kRawOff);
#pragma unroll
    for (int q = 0; q < kRounds / 2; q++) raw32[64 * q + lane] = d.v[q];
    __builtin_amdgcn_wave_barrier();
    const int16_t *raw = st + kRawOff;
    const uint32_t st_addr = lds_offset(st);
    uint32_t run = shift;
#pragma unroll
    for (int r = 0; r < kRounds; r++) {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)m, r), hi = (uint32_t)__builtin_amdgcn_readlane((int)(m >> 32), r);
        const uint32_t rank = (uint32_t)__builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, run));
        const uint32_t v = (uint16_t)raw[64 * r + lane];
        const uint32_t addr = st_addr + 2 * rank;
        const uint64_t mask = ((uint64_t)hi << 32) | lo;
        uint64_t saved;
        asm volatile("s_mov_b64 %[sv], exec\n\ts_mov_b64 exec, %[m]\n\tds_write_b16 %[a], %[v]\n\ts_mov_b64 exec, %[sv]"
                     : [sv] "=&s"(saved) : [m] "s"(mask), [a] "v"(addr), [v] "v"(v) : "memory");
        run += (uint32_t)__builtin_popcountll(mask);
    }
    __builtin_amdgcn_wave_barrier();
    return run - shift;
}


template <>
struct Placer<EmitI16> {
    typedef TileDraws Draws;
    __device__ static Draws load(const int16_t *rec) { return load_tile_i16(rec); }
    __device__ static void place(const NpJob &job, const Draws &d, const int16_t *, uint64_t m, long long prefix, int16_t *st, uint32_t &)
{
    const uint32_t shift = (uint32_t)prefix & 1u;
    uint32_t total = stage_tile_lds(d, m, shift, st);
    if (prefix + total > job.n) total = (uint32_t)(job.n - prefix);
    finish(job, total, shift, prefix, st);
}
    // slots [shift, shift + total) of `st` hold the tile's samples: whole dwords to dst[prefix ...]
    __device__ static void finish(const NpJob &job, uint32_t total, uint32_t shift, long long prefix, int16_t *st)
{
    const uint32_t lane = (uint32_t)__lane_id();
    int16_t VKX_GLOBAL *dst = (int16_t VKX_GLOBAL *)job.dst + (prefix - shift);
    const uint32_t end = shift + total;                // slots [shift, end) are valid
    const uint32_t full_hi = end >> 1;                 // dwords [shift, full_hi) are whole
    uint32_t VKX_GLOBAL *dst32 = (uint32_t VKX_GLOBAL *)dst + lane;
    const uint32_t *st32 = (const uint32_t *)st + lane;
    if (lane >= shift && lane < full_hi) dst32[0] = st32[0];
#pragma unroll
    for (uint32_t i = 1; i < kTile / 128; i++)
        if (lane + 64 * i < full_hi) dst32[64 * i] = st32[64 * i];
    if (lane == 0) {
        if (shift && end > 1) dst[1] = st[1];
        if ((end & 1) && end - 1 >= shift) dst[end - 1] = st[end - 1];
    }
    __builtin_amdgcn_wave_barrier();
}
};

template <>
struct Placer<EmitAddU8> {
    typedef TileDraws Draws;
    __device__ static Draws load(const int16_t *rec) { return load_tile_i16(rec); }
    __device__ static void place(const NpJob &job, const Draws &d, const int16_t *, uint64_t m, long long prefix, int16_t *st, uint32_t &)
{
    const uint32_t shift = (uint32_t)prefix & 3u;
    // the pixels the tile's samples go onto are requested before the samples are compacted: their latency hides behind
    // the staging (every whole dword of the plane that the tile can reach; what the count leaves unused is dropped)
    const uint32_t lane = (uint32_t)__lane_id();
    const long long first = prefix - shift;
    const uint32_t VKX_GLOBAL *src32 = (const uint32_t VKX_GLOBAL *)((const uint8_t VKX_GLOBAL *)job.src + first) + lane;
    uint32_t px[kTile / 256];
#pragma unroll
    for (uint32_t i = 0; i < kTile / 256; i++) {      // shift + total <= 1027: whole dwords 0 .. 255
        px[i] = 0;
        if (first + 4 * (long long)(lane + 64 * i) + 3 < job.n) px[i] = src32[64 * i];
    }
    uint32_t total = stage_tile_lds(d, m, shift, st);
    if (prefix + total > job.n) total = (uint32_t)(job.n - prefix);
    finish_px(job, total, shift, prefix, st, px);
}
    __device__ static void finish(const NpJob &job, uint32_t total, uint32_t shift, long long prefix, int16_t *st)
{
    const uint32_t lane = (uint32_t)__lane_id();
    const long long first = prefix - shift;
    const uint32_t VKX_GLOBAL *src32 = (const uint32_t VKX_GLOBAL *)((const uint8_t VKX_GLOBAL *)job.src + first) + lane;
    uint32_t px[kTile / 256];
#pragma unroll
    for (uint32_t i = 0; i < kTile / 256; i++) {      // shift + total <= 1027: whole dwords 0 .. 255
        px[i] = 0;
        if (first + 4 * (long long)(lane + 64 * i) + 3 < job.n) px[i] = src32[64 * i];
    }
    finish_px(job, total, shift, prefix, st, px);
}
    __device__ static void finish_px(const NpJob &job, uint32_t total, uint32_t shift, long long prefix, int16_t *st,
                                     const uint32_t (&pxs)[kTile / 256])
{
    typedef short pk16 __attribute__((ext_vector_type(2)));
    const uint32_t lane = (uint32_t)__lane_id();
    const uint8_t VKX_GLOBAL *src = (const uint8_t VKX_GLOBAL *)job.src + (prefix - shift);
    uint8_t VKX_GLOBAL *dst = (uint8_t VKX_GLOBAL *)job.dst + (prefix - shift);
    const uint32_t end = shift + total;
    const uint32_t full_lo = (shift + 3) >> 2, full_hi = end >> 2;      // dwords [full_lo, full_hi) are whole
    uint32_t VKX_GLOBAL *dst32 = (uint32_t VKX_GLOBAL *)dst + lane;
    const uint2 *st64 = (const uint2 *)st + lane;
    auto whole = [&](uint32_t i) {
        const uint32_t px = pxs[i];
        const uint2 nz = st64[64 * i];
        // bytes 0 1 | 2 3 of the pixel dword as int16 pairs, + noise, clamp, repack
        pk16 lo = __builtin_bit_cast(pk16, __builtin_amdgcn_perm(0u, px, 0x0c010c00u));
        pk16 hi = __builtin_bit_cast(pk16, __builtin_amdgcn_perm(0u, px, 0x0c030c02u));
        lo += __builtin_bit_cast(pk16, nz.x);
        hi += __builtin_bit_cast(pk16, nz.y);
        const pk16 zero = {0, 0}, top = {255, 255};
        lo = __builtin_elementwise_min(__builtin_elementwise_max(lo, zero), top);
        hi = __builtin_elementwise_min(__builtin_elementwise_max(hi, zero), top);
        dst32[64 * i] = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, hi), __builtin_bit_cast(uint32_t, lo), 0x06040200u);
    };
    if (lane >= full_lo && lane < full_hi) whole(0);
#pragma unroll
    for (uint32_t i = 1; i < kTile / 256; i++)
        if (lane + 64 * i < full_hi) whole(i);
    // the bytes of the partial dwords at both ends
    if (lane < 8) {
        const uint32_t e = lane < 4 ? lane : 4 * full_hi + (lane - 4);
        const bool mine = lane < 4 ? (shift > 0 && e >= shift && e < end) : (e < end && !(full_hi == 0 && shift > 0));
        if (mine) dst[e] = (uint8_t)vkd::clamp_u8((int16_t)((int)src[e] + (int)st[e]));
    }
    __builtin_amdgcn_wave_barrier();
}
};

// Pass 2: the recorded draws of a tile go to their final index (k_np_place, one wavefront per tile, nothing but the
// tile's records, masks and plan to read).  Tiles whose carry-in does not merge into the recorded chain (~2e-4 of them),
// the tile holding a job's last sample (its draw count is reported) and, for emitters that check a tail sample against
// its pixel, tiles with a tail sample are left to k_np_place_walk, which walks them again from the generator state.
__device__ __forceinline__ bool tile_needs_walk(const NpJob &job, const TilePlan &p, uint32_t out0, bool check_at_store)
{
    return (p.c_in & kIrregular) || (long long)p.prefix + p.count >= job.n || (check_at_store && (out0 >> 31));
}

template <class Emit>
__global__ void __launch_bounds__(256) k_np_place(const NpJob *__restrict__ jobs, int n_jobs, long long total_tiles,
                                                  const TileInfo *__restrict__ info, const TilePlan *__restrict__ plan,
                                                  const typename Emit::Store *__restrict__ rec_val, const uint64_t *__restrict__ rec_mask,
                                                  vkx_np_result *__restrict__ results)
{
    __shared__ __attribute__((aligned(16))) typename Emit::Store stage[4][kTile + 64];
    const int lane = __lane_id();
    const long long tile = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= total_tiles) return;
    // everything the tile reads is requested before the plan decides (one memory round trip per tile)
    const TilePlan p = plan[tile];
    uint64_t m = lane < kRounds ? ((const uint64_t VKX_GLOBAL *)rec_mask)[tile * kRounds + lane] : 0ull;
    const typename Placer<Emit>::Draws draws = Placer<Emit>::load(rec_val + tile * kTile);
    const uint32_t out0 = Emit::kCheckAtStore ? info[tile].out0 : 0u;
    const int j = job_of_tile(jobs, n_jobs, tile);
    const NpJob &job = jobs[j];
    const long long prefix = (long long)rfl64(p.prefix);
    if (prefix >= job.n) return;
    TilePlan pu;
    pu.prefix = (unsigned long long)prefix;
    pu.c_in = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.c_in);
    pu.count = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.count);
    if (tile_needs_walk(job, pu, (uint32_t)__builtin_amdgcn_readfirstlane((int)out0), Emit::kCheckAtStore)) return;
    uint32_t flags = 0;
    if (lane == 0) m &= ~((1ull << pu.c_in) - 1);     // c_in < 64 is a start of the recorded chain: the draws before it belong to the previous tile
    Placer<Emit>::place(job, draws, rec_val + tile * kTile, m, prefix, stage[threadIdx.x >> 6], flags);
    if (flags) atomicOr(&results[j].flags, flags);
}

// Pass 2 of the compact kinds, one wavefront per tile: the samples of slot elements [skip, skip + count) go to their final
// index -- a copy (VKX_NP_NORMAL_I16) or clip(uint8 + int16) in place (VKX_NP_NORMAL_ADD_U8), four elements per lane and step
// (an unaligned 8-byte record load; the destination group is aligned).  `skip` = the samples the recorded chain emitted
// before the tile's true carry-in.
template <bool ADD>
__device__ __forceinline__ void apply_tile(const int16_t VKX_GLOBAL *rec /* the tile's first valid sample */, long long prefix, uint32_t count,
                                           const uint8_t *src, void *dst)
{
    typedef short pk16 __attribute__((ext_vector_type(2)));
    typedef unsigned long long u64_u2 __attribute__((aligned(2)));
    const uint32_t lane = (uint32_t)__lane_id();
    const uint32_t head = min(count, (uint32_t)(-prefix) & 3u);
    const uint32_t body = (count - head) >> 2, tail = (count - head) & 3u;
    auto one = [&](uint32_t e) {
        if (ADD) {
            const int v = (int16_t)((int)((const uint8_t VKX_GLOBAL *)src)[prefix + e] + (int)rec[e]);
            ((uint8_t VKX_GLOBAL *)dst)[prefix + e] = (uint8_t)vkd::clamp_u8(v);
        } else {
            ((int16_t VKX_GLOBAL *)dst)[prefix + e] = rec[e];
        }
    };
#pragma unroll 4
    for (uint32_t i = 0; i < (uint32_t)kTile / 256; i++) {
        if (64u * i >= body) break;
        const uint32_t g = lane + 64u * i;
        if (g < body) {
            const uint32_t e = head + 4u * g;
            const unsigned long long nz = *(const u64_u2 VKX_GLOBAL *)(rec + e);
            if (ADD) {
                const uint32_t px = *(const uint32_t VKX_GLOBAL *)((const uint8_t VKX_GLOBAL *)src + prefix + e);
                pk16 lo = __builtin_bit_cast(pk16, __builtin_amdgcn_perm(0u, px, 0x0c010c00u));
                pk16 hi = __builtin_bit_cast(pk16, __builtin_amdgcn_perm(0u, px, 0x0c030c02u));
                lo += __builtin_bit_cast(pk16, (uint32_t)nz);
                hi += __builtin_bit_cast(pk16, (uint32_t)(nz >> 32));
                const pk16 zero = {0, 0}, top = {255, 255};
                lo = __builtin_elementwise_min(__builtin_elementwise_max(lo, zero), top);
                hi = __builtin_elementwise_min(__builtin_elementwise_max(hi, zero), top);
                *(uint32_t VKX_GLOBAL *)((uint8_t VKX_GLOBAL *)dst + prefix + e) =
                    __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, hi), __builtin_bit_cast(uint32_t, lo), 0x06040200u);
            } else {
                *(unsigned long long VKX_GLOBAL *)((int16_t VKX_GLOBAL *)dst + prefix + e) = nz;
            }
        }
    }
    if (lane < 8) {
        const uint32_t k = lane & 3u;
        if (lane < 4 ? k < head : k < tail) one(lane < 4 ? k : head + 4u * body + k);
    }
}

template <bool ADD>
__global__ void __launch_bounds__(256) k_np_apply(const NpJob *__restrict__ jobs, int n_jobs, long long total_tiles,
                                                  const TileInfo *__restrict__ info, const TilePlan *__restrict__ plan)
{
    const long long tile = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= total_tiles) return;
    const TilePlan p = plan[tile];
    const uint64_t emit0 = info[tile].emit0;
    const NpJob &job = jobs[job_of_tile(jobs, n_jobs, tile)];
    const long long prefix = (long long)rfl64(p.prefix);
    if (prefix >= job.n) return;
    TilePlan pu;
    pu.prefix = (unsigned long long)prefix;
    pu.c_in = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.c_in);
    pu.count = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.count);
    if (tile_needs_walk(job, pu, 0u, false)) return;
    const uint32_t skip = (uint32_t)__builtin_popcountll(rfl64(emit0) & ((1ull << pu.c_in) - 1));
    // prefix + count < n here (the tile with the last sample is walked)
    apply_tile<ADD>((const int16_t VKX_GLOBAL *)job.rec + (tile - job.tile_base) * kSlot + skip, prefix, pu.count, job.src, job.dst);
}

// A finished tile buffer (VKX_NP_NORMAL_TILES) as the int16 plane it stands for: vkx_np_tiles_expand_dev.
__global__ void __launch_bounds__(256) k_np_tiles_expand(const uint2 *__restrict__ table, const int16_t *__restrict__ slots, int n_tiles,
                                                         long long n, int16_t *__restrict__ dst)
{
    const int t = (int)(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (t >= n_tiles) return;
    const uint2 a = table[t], b = table[t + 1];
    const long long prefix = (long long)__builtin_amdgcn_readfirstlane((int)a.x) & 0xffffffffLL;
    const uint32_t next = (uint32_t)__builtin_amdgcn_readfirstlane((int)b.x), skip = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.y);
    if (prefix >= n) return;
    uint32_t count = next - (uint32_t)prefix;
    if (prefix + count > n) count = (uint32_t)(n - prefix);
    apply_tile<false>((const int16_t VKX_GLOBAL *)slots + (long long)t * kSlot + skip, prefix, count, nullptr, dst);
}

// VKX_NP_NORMAL_TILES, after the walk: one lane per tile writes the tile's table entry (index of its first sample, first valid
// slot element) and lends the tile the first two samples of its successor, so that the three samples of a pixel never
// straddle two slots; the job's last tile also writes the sentinel entry and the header.
__global__ void __launch_bounds__(256) k_np_tiles_finish(const NpJob *__restrict__ jobs, const TileInfo *__restrict__ info,
                                                         const TilePlan *__restrict__ plan)
{
    const NpJob &job = jobs[blockIdx.y];           // grid = (tiles of the longest stream / 256, streams)
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= job.n_tiles) return;
    const long long tile = job.tile_base + t;
    auto skip_of = [&](long long tl, const TilePlan &p) -> uint32_t {
        if ((long long)p.prefix < job.n && tile_needs_walk(job, p, 0u, false)) return 0u;     // rewritten from its true carry-in
        if (p.c_in & kIrregular) return 0u;                                                   // (beyond the samples wanted)
        return (uint32_t)__builtin_popcountll(info[tl].emit0 & ((1ull << p.c_in) - 1));
    };
    const TilePlan p = plan[tile];
    const uint32_t skip = skip_of(tile, p);
    job.table[t] = make_uint2((uint32_t)p.prefix, skip);
    int16_t *rec = job.rec + t * kSlot;
    if (t + 1 < job.n_tiles) {
        const TilePlan pn = plan[tile + 1];
        const int16_t *nxt = rec + kSlot + skip_of(tile + 1, pn);
        rec[skip + p.count] = nxt[0];
        rec[skip + p.count + 1] = nxt[1];
    } else {
        job.table[t + 1] = make_uint2((uint32_t)(p.prefix + p.count), 0u);
        uint32_t *header = (uint32_t *)job.table - 4;
        header[0] = (uint32_t)job.n_tiles;
        header[1] = (uint32_t)kSlot;
        *(unsigned long long *)(header + 2) = p.prefix + p.count;
    }
}

// (one wavefront per workgroup: the kEmit workspace of a float64 emitter is 26 KB)
// TILES: the walked tiles rewrite their slot from its first element (their table entry says skip = 0)
template <class Emit, bool TILES = false, int STEP = 64>
__global__ void __launch_bounds__(64) k_np_place_walk(const NpJob *__restrict__ jobs, int n_jobs, long long total_tiles,
                                                       const uint64_t *__restrict__ states, const TileInfo *__restrict__ info,
                                                       const TilePlan *__restrict__ plan, vkx_np_result *__restrict__ results,
                                                       unsigned *__restrict__ done, vkx_np_result *__restrict__ results_host,
                                                       const NpTabs *__restrict__ tabs)
{
    __shared__ uint4 zig[256];
    __shared__ double fi[256];
    __shared__ __attribute__((aligned(16))) WaveWork<typename Emit::Store, true> work[1];
    load_tables(zig, fi, tabs);
    const JumpTabs &g_jump = tabs->jump;
    const int lane = __lane_id();
    const long long n_waves = (long long)gridDim.x;
    // STEP tiles per step: a lane looks at one plan, the wavefront then walks the few tiles that asked for it (STEP = 64 where walks are
    // rare; the speckle records are walked wherever a tile holds a tail draw -- 43 % of the 3 072-draw tiles --: 4 tiles per wavefront)
    for (long long t0 = (long long)blockIdx.x * STEP; t0 < total_tiles; t0 += n_waves * STEP) {
        const long long mine = t0 + lane;
        bool want = false;
        if (lane < STEP && mine < total_tiles) {
            const NpJob &jb = jobs[job_of_tile(jobs, n_jobs, mine)];
            const TilePlan p = plan[mine];
            want = (long long)p.prefix < jb.n && tile_needs_walk(jb, p, info[mine].out0, Emit::kCheckAtStore);
        }
        uint64_t todo = __ballot(want);
        while (todo) {
            const int b = __builtin_ctzll(todo);
            todo &= todo - 1;
            const long long tile = t0 + b;
            const int j = job_of_tile(jobs, n_jobs, tile);
            const NpJob &job = jobs[j];
            const TilePlan p = plan[tile];
            uint32_t count, carry, flags = 0;
            uint64_t start0, emit0;
            bool ht;
            void *dst = job.dst;
            if (TILES) dst = (void *)((intptr_t)(job.rec + (tile - job.tile_base) * kSlot) - 2 * (intptr_t)p.prefix);
            walk_tile<Emit, kEmit>(job, g_jump, zig, fi, work[0], mk128(&states[2 * tile]), p.c_in & ~kIrregular,
                                   (long long)p.prefix, (tile - job.tile_base) * kTile, nullptr, nullptr, dst, count, carry, start0, emit0, ht,
                                   flags, &results[j].draws);
            if (flags) atomicOr(&results[j].flags, flags);
        }
    }
    np_results_out(done, results, results_host, n_jobs);
}

// ---- uniform doubles: one draw per element (Generator.random / Generator.choice with p) -----------------------------
// impulse_noise: selector = #{k : cdf[k] <= u} per PIXEL (0 keep, 1 salt, 2 pepper), applied to all cn channels.
__global__ void __launch_bounds__(256) k_np_choice_impulse(const NpJob *__restrict__ jobs, int n_jobs, long long total_tiles,
                                                           const uint64_t *__restrict__ states, vkx_np_result *__restrict__ results,
                                                           unsigned *__restrict__ done, vkx_np_result *__restrict__ results_host,
                                                           const NpTabs *__restrict__ tabs)
{
    const JumpTabs &g_jump = tabs->jump;
    const long long n_waves = (long long)gridDim.x * 4;
    const int lane = __lane_id();
    for (long long tile = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); tile < total_tiles; tile += n_waves) {
        const int jb = job_of_tile(jobs, n_jobs, tile);
        const NpJob &job = jobs[jb];
        if (tile == job.tile_base && lane == 0) {
            __hip_atomic_store(&results[jb].draws, (unsigned long long)job.n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&results[jb].samples, (unsigned long long)job.n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const u128 inc = mk128(job.inc), a64 = mk128(g_jump.a64), c64 = mk128(job.c64);
        u128 s = mk128(&g_jump.lane[lane][0]) * mk128(&states[2 * tile]) + mk128(&g_jump.lane[lane][2]) * inc;
        const long long e0 = (tile - job.tile_base) * kTile + lane;
        const int cn = job.cn;
#pragma unroll 1
        for (int r = 0; r < kRounds; r++) {
            const long long e = e0 + 64 * r;
            if (e - lane >= job.n) break;
            const double u = u2dbl(pcg_out(s));
            if (e < job.n) {
                const int sel = (job.cdf[0] <= u) + (job.cdf[1] <= u) + (job.cdf[2] <= u);
                uint8_t *d = (uint8_t *)job.dst + e * cn;
                if (job.kind == VKX_NP_CHOICE3_U8) {
                    d[0] = (uint8_t)sel;
                } else {
                    const uint8_t *p = job.src + e * cn;
                    for (int k = 0; k < cn; k++) d[k] = sel == 1 ? 255 : (sel == 2 ? 0 : p[k]);
                }
            }
            s = a64 * s + c64;
        }
    }
    np_results_out(done, results, results_host, n_jobs);
}

} // namespace

static int np_tables(vkx_ctx *ctx, const NpTabs **out)
{
    if (!ctx->np_tabs.ptr) {
        build_jump_tabs();
        int rc = vkx_scratch_reserve(ctx, &ctx->np_tabs, sizeof(NpTabs));
        if (rc) return rc;
        void *ring = nullptr;
        if ((rc = vkx_desc_ring_take(ctx, sizeof(NpTabs), &ring))) return rc;
        NpTabs *t = (NpTabs *)ring;
        t->jump = g_jump_host;
        memcpy(t->ki, kNpZigK, 2048);
        memcpy(t->wi, kNpZigW, 2048);
        memcpy(t->fi, kNpZigF, 2048);
        vkx_device_guard guard(ctx);
        VKX_HIP(hipMemcpyAsync(ctx->np_tabs.ptr, t, sizeof(NpTabs), hipMemcpyHostToDevice, ctx->stream));
    }
    *out = (const NpTabs *)ctx->np_tabs.ptr;
    return VKX_OK;
}

static long long np_tiles_for_n(long long n, bool uniform)
{
    // raw draws to provision: numpy's ziggurat takes 1.022025 per sample (2e8 samples of the CPU restatement; 1.02198 .. 1.02205 per 25 M), the count
    // of a stream of n samples scatters with sigma = 0.146 sqrt(n) (557 at 14.5 M): n / 45 = 0.02222 n plus 4 096 is 12 sigma at 14.5 M samples and more
    // everywhere else.  (Through round 4: n / 32 + 2 048 -- 0.9 % more tiles for the draw pass to walk.)  A stream that runs short all the same raises
    // VKX_NP_SHORT and the caller draws with numpy.
    const long long draws = uniform ? n : n + n / 45 + 4096;
    return (draws + kTile - 1) / kTile;
}
static long long np_tiles_for(const vkx_np_job &j, bool uniform) { return np_tiles_for_n(j.n, uniform); }

// The tile buffer of a VKX_NP_NORMAL_TILES job of n samples: 16-byte header, the table, the slots.
vkx_np_tiles_shape vkx_np_tiles_shape_of(long long n)
{
    vkx_np_tiles_shape s;
    s.n_tiles = np_tiles_for_n(n, false);
    s.slot_elems = kSlot;
    s.table_offset = 16;
    s.slots_offset = (16 + 8 * ((size_t)s.n_tiles + 1) + 255) & ~(size_t)255;
    s.bytes = s.slots_offset + (size_t)s.n_tiles * kSlot * 2;
    s.samples_per_tile = (double)kTile * 0.97850;      // E[samples per raw draw] of numpy's ziggurat: 0.97850 (measured over 3e6 samples)
    return s;
}

VKX_EXPORT int vkx_np_tiles_layout(int64_t n, int64_t *n_tiles, int64_t *slot_elems, int64_t *table_offset, int64_t *slots_offset,
                                   int64_t *bytes)
{
    VKX_REQUIRE(n >= 1 && n <= 0x7fffffffLL, "1 .. 2^31 - 1 samples");
    const vkx_np_tiles_shape s = vkx_np_tiles_shape_of(n);
    if (n_tiles) *n_tiles = s.n_tiles;
    if (slot_elems) *slot_elems = s.slot_elems;
    if (table_offset) *table_offset = (int64_t)s.table_offset;
    if (slots_offset) *slots_offset = (int64_t)s.slots_offset;
    if (bytes) *bytes = (int64_t)s.bytes;
    return VKX_OK;
}

// VKX_NP_COMPACT=0: the int16 kinds through the raw-draw records and the placement pass of round 3 (A/B).
static bool np_compact_records(int kind)
{
    static const bool on = [] { const char *e = getenv("VKX_NP_COMPACT"); return !(e && e[0] == '0'); }();
    return kind == VKX_NP_NORMAL_TILES || (on && (kind == VKX_NP_NORMAL_I16 || kind == VKX_NP_NORMAL_ADD_U8));
}

// One chunk of a call: its tile arrays in one of the two scratch slots, its jobs, its slice of the results.
struct NpChunk {
    const vkx_np_job *jobs;
    int n_jobs, kind, max_tiles, slot;
    bool uniform, inline_jobs;
    long long total_tiles;
    vkx_np_result *results_host, *res_mapped;
    unsigned char *base;
    size_t o_states, o_info, o_plan, o_rmask, o_rval, o_jobs, o_results;
    NpJobPack pack;
    const NpTabs *tabs;
};

static int np_chunk_prepare(vkx_ctx *ctx, NpChunk &c)
{
    const bool uniform = c.uniform;
    c.total_tiles = 0;
    c.max_tiles = 0;
    for (int i = 0; i < c.n_jobs; i++) {
        const long long tiles = np_tiles_for(c.jobs[i], uniform);
        c.max_tiles = std::max<int>(c.max_tiles, (int)tiles);
        c.total_tiles += tiles;
    }
    size_t off = 0;
    auto take = [&off](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    c.o_states = take((size_t)c.total_tiles * 16);
    c.o_info = take(uniform ? 0 : (size_t)c.total_tiles * sizeof(TileInfo));
    c.o_plan = take(uniform ? 0 : (size_t)c.total_tiles * sizeof(TilePlan));
    const size_t val_bytes = c.kind == VKX_NP_SPECKLE_U8 ? 8 : 2;
    const bool compact = !uniform && np_compact_records(c.kind);
    c.o_rmask = take(uniform || compact ? 0 : (size_t)c.total_tiles * kRounds * 8);
    c.o_rval = take(uniform || c.kind == VKX_NP_NORMAL_TILES ? 0 : (size_t)c.total_tiles * (compact ? kSlot : kTile) * val_bytes);
    c.o_jobs = take((size_t)c.n_jobs * sizeof(NpJob));
    c.o_results = take((size_t)c.n_jobs * sizeof(vkx_np_result) + sizeof(unsigned));    // + the completion counter
    int rc = vkx_scratch_reserve(ctx, &ctx->np_work[c.slot], off);
    if (rc) return rc;
    c.base = (unsigned char *)ctx->np_work[c.slot].ptr;

    // the device forms of the jobs: in the kernel arguments of the first kernel (small calls) or through the ctx ring
    c.inline_jobs = c.n_jobs <= kInlineJobs;
    NpJob *hj = c.pack.jobs;
    if (!c.inline_jobs) {
        void *ring = nullptr;
        if ((rc = vkx_desc_ring_take(ctx, (size_t)c.n_jobs * sizeof(NpJob), &ring))) return rc;
        hj = (NpJob *)ring;
    }
    const u128 g64 = ((u128)g_jump_host.g64[1] << 64) | g_jump_host.g64[0];
    const u128 g128 = ((u128)g_jump_host.g128[1] << 64) | g_jump_host.g128[0];
    long long tile_base = 0;
    for (int i = 0; i < c.n_jobs; i++) {
        const vkx_np_job &j = c.jobs[i];
        NpJob &d = hj[i];
        d.state[0] = j.state[0]; d.state[1] = j.state[1];
        d.inc[0] = j.inc[0]; d.inc[1] = j.inc[1];
        const u128 c64 = (((u128)j.inc[1] << 64) | j.inc[0]) * g64;
        d.c64[0] = (uint64_t)c64; d.c64[1] = (uint64_t)(c64 >> 64);
        const u128 c128 = (((u128)j.inc[1] << 64) | j.inc[0]) * g128;
        d.c128[0] = (uint64_t)c128; d.c128[1] = (uint64_t)(c128 >> 64);
        d.n = j.n;
        d.tile_base = tile_base;
        d.n_tiles = (int)np_tiles_for(j, uniform);
        tile_base += d.n_tiles;
        d.kind = j.kind & 0xff; d.cn = j.cn;
        d.margin = (j.kind & VKX_NP_DEBUG_WIDE_MARGIN) ? 1.0 : 0x1p-42;
        d.loc = 0.0; d.scale = j.scale;
        d.cdf[0] = j.cdf[0]; d.cdf[1] = j.cdf[1]; d.cdf[2] = j.cdf[2];
        d.src = (const uint8_t *)j.src;
        d.dst = j.dst;
        d.rec = nullptr;
        d.table = nullptr;
        if (c.kind == VKX_NP_NORMAL_TILES) {
            const vkx_np_tiles_shape shape = vkx_np_tiles_shape_of(j.n);
            d.rec = (int16_t *)((unsigned char *)j.dst + shape.slots_offset);
            d.table = (uint2 *)((unsigned char *)j.dst + shape.table_offset);
        } else if (compact) {
            d.rec = (int16_t *)(c.base + c.o_rval) + d.tile_base * kSlot;
        }
    }
    if (!c.inline_jobs && (rc = vkx_small_to_device(ctx, c.base + c.o_jobs, hj, (size_t)c.n_jobs * sizeof(NpJob)))) return rc;
    // page-locked results: the chunk's last kernel writes them through the mapping
    c.res_mapped = nullptr;
    if (hipHostGetDevicePointer((void **)&c.res_mapped, c.results_host, 0) != hipSuccess) {
        (void)hipGetLastError();
        c.res_mapped = nullptr;
    }
    return VKX_OK;
}

// tile states + the draw pass (VALU bound); the uniform kinds are complete after it.  On ctx->stream.
static int np_chunk_front(vkx_ctx *ctx, NpChunk &c)
{
    unsigned char *base = c.base;
    const NpJob *dj = (const NpJob *)(base + c.o_jobs);
    uint64_t *states = (uint64_t *)(base + c.o_states);
    TileInfo *info = (TileInfo *)(base + c.o_info);
    vkx_np_result *res = (vkx_np_result *)(base + c.o_results);
    unsigned *done = (unsigned *)(res + c.n_jobs);
    const long long total_tiles = c.total_tiles;
    const int n_jobs = c.n_jobs, kind = c.kind;
    const NpTabs *tabs = c.tabs;
    // workgroups of a few tiles per wavefront: a chunk's draw shares the device with the placement pass of the chunk before it
    // (2 048 persistent workgroups -- two generations of the 4 096 wavefronts the device holds -- measured 8.3 ms per 256
    // planes, workgroups of 4 tiles per wavefront 7.5 ms: the dispatcher balances them, and the placement pass gets slots)
    // (round 5, swept again under the joint call: 1 / 2 / 4 tiles per wavefront draw ONE 2048^2 plane in 0.050 / 0.064 / 0.083 ms, four planes in 0.132 /
    //  0.139 / 0.172, sixteen in 0.475 / 0.466 / 0.471, and the bench's chunk of 128 planes -- with the step's small kernels under it -- in 3.79 / 3.67 / 3.63:
    //  the step 16.77 / 16.52 / 16.41 ms; 8 and 12: 16.55 / 16.60)
    static const int tpw_env = [] { const char *e = getenv("VKX_NP_TPW"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 0; }();
    const int tiles_per_wave = tpw_env ? tpw_env : (total_tiles >= 160000 ? 4 : total_tiles >= 40000 ? 2 : 1);
    const unsigned wg = (unsigned)((total_tiles + 4 * tiles_per_wave - 1) / (4 * tiles_per_wave));
    {
        VKX_TIMED(ctx, "k_np_tile_states");
        if (c.inline_jobs)
            k_np_tile_states<true><<<vkx_blocks((size_t)total_tiles, 256), 256, 0, ctx->stream>>>(dj, c.pack, n_jobs, total_tiles, states, (NpJob *)(base + c.o_jobs), res, done, tabs);
        else
            k_np_tile_states<false><<<vkx_blocks((size_t)total_tiles, 256), 256, 0, ctx->stream>>>(dj, c.pack, n_jobs, total_tiles, states, (NpJob *)(base + c.o_jobs), res, done, tabs);
        VKX_LAUNCH_CHECK();
    }
    if (c.uniform) {
        VKX_TIMED(ctx, "k_np_choice_impulse");
        k_np_choice_impulse<<<wg, 256, 0, ctx->stream>>>(dj, n_jobs, total_tiles, states, res, done, c.res_mapped, tabs);
        VKX_LAUNCH_CHECK();
        return VKX_OK;
    }
    if (np_compact_records(kind)) {
        // (the emitter only decides how a draw becomes an int16 step: the same for the three int16 kinds)
        VKX_TIMED_MAJOR(ctx, "k_np_draw");
        const unsigned wgc = (unsigned)((total_tiles + (long long)kDrawWaves * tiles_per_wave - 1) / ((long long)kDrawWaves * tiles_per_wave));
        k_np_draw_compact<EmitI16><<<wgc, 64 * kDrawWaves, 0, ctx->stream>>>(dj, n_jobs, total_tiles, states, info, res, tabs);
        VKX_LAUNCH_CHECK();
        return VKX_OK;
    }
    uint64_t *rmask = (uint64_t *)(base + c.o_rmask);
    void *rval = base + c.o_rval;
    VKX_TIMED_MAJOR(ctx, "k_np_draw");
    if (kind == VKX_NP_SPECKLE_U8)
        k_np_draw<EmitSpeckle><<<wg, 256, 0, ctx->stream>>>(dj, n_jobs, total_tiles, states, info, (double *)rval, rmask, res, tabs);
    else if (kind == VKX_NP_NORMAL_ADD_U8)
        k_np_draw<EmitAddU8><<<wg, 256, 0, ctx->stream>>>(dj, n_jobs, total_tiles, states, info, (int16_t *)rval, rmask, res, tabs);
    else
        k_np_draw<EmitI16><<<wg, 256, 0, ctx->stream>>>(dj, n_jobs, total_tiles, states, info, (int16_t *)rval, rmask, res, tabs);
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}

// resolve + place + walk (memory / latency bound) and the results.  On ctx->stream.
static int np_chunk_back(vkx_ctx *ctx, NpChunk &c)
{
    if (c.uniform) {
        if (!c.res_mapped)
            VKX_HIP(hipMemcpyAsync(c.results_host, c.base + c.o_results, (size_t)c.n_jobs * sizeof(vkx_np_result), hipMemcpyDeviceToHost, ctx->stream));
        return VKX_OK;
    }
    unsigned char *base = c.base;
    const NpJob *dj = (const NpJob *)(base + c.o_jobs);
    uint64_t *states = (uint64_t *)(base + c.o_states);
    TileInfo *info = (TileInfo *)(base + c.o_info);
    TilePlan *plan = (TilePlan *)(base + c.o_plan);
    vkx_np_result *res = (vkx_np_result *)(base + c.o_results);
    unsigned *done = (unsigned *)(res + c.n_jobs);
    uint64_t *rmask = (uint64_t *)(base + c.o_rmask);
    void *rval = base + c.o_rval;
    const long long total_tiles = c.total_tiles;
    const int n_jobs = c.n_jobs, kind = c.kind;
    const NpTabs *tabs = c.tabs;
    {
        VKX_TIMED(ctx, "k_np_resolve");
        const size_t bits = (((size_t)c.max_tiles + 63) / 64) * 8;
        k_np_resolve<<<n_jobs, 1024, bits, ctx->stream>>>(dj, states, info, plan, res, tabs);
        VKX_LAUNCH_CHECK();
    }
    if (np_compact_records(kind)) {
        const unsigned pg = vkx_blocks((size_t)total_tiles, 4);
        if (kind != VKX_NP_NORMAL_TILES) {
            VKX_TIMED(ctx, "k_np_apply");
            if (kind == VKX_NP_NORMAL_ADD_U8) k_np_apply<true><<<pg, 256, 0, ctx->stream>>>(dj, n_jobs, total_tiles, info, plan);
            else k_np_apply<false><<<pg, 256, 0, ctx->stream>>>(dj, n_jobs, total_tiles, info, plan);
            VKX_LAUNCH_CHECK();
        }
    } else {
        VKX_TIMED(ctx, "k_np_place");
        const unsigned pg = vkx_blocks((size_t)total_tiles, 4);
        if (kind == VKX_NP_SPECKLE_U8)
            k_np_place<EmitSpeckle><<<pg, 256, 0, ctx->stream>>>(dj, n_jobs, total_tiles, info, plan, (const double *)rval, rmask, res);
        else if (kind == VKX_NP_NORMAL_ADD_U8)
            k_np_place<EmitAddU8><<<pg, 256, 0, ctx->stream>>>(dj, n_jobs, total_tiles, info, plan, (const int16_t *)rval, rmask, res);
        else
            k_np_place<EmitI16><<<pg, 256, 0, ctx->stream>>>(dj, n_jobs, total_tiles, info, plan, (const int16_t *)rval, rmask, res);
        VKX_LAUNCH_CHECK();
    }
    {
        VKX_TIMED(ctx, "k_np_place_walk");
        const unsigned wg2 = (unsigned)std::min<long long>((total_tiles + 63) / 64, 256 * 16);
        if (kind == VKX_NP_SPECKLE_U8)
            k_np_place_walk<EmitSpeckle, false, 4><<<(unsigned)std::min<long long>((total_tiles + 3) / 4, 256 * 16), 64, 0, ctx->stream>>>(
                dj, n_jobs, total_tiles, states, info, plan, res, done, c.res_mapped, tabs);
        else if (kind == VKX_NP_NORMAL_ADD_U8)
            k_np_place_walk<EmitAddU8><<<wg2, 64, 0, ctx->stream>>>(dj, n_jobs, total_tiles, states, info, plan, res, done, c.res_mapped, tabs);
        else if (kind == VKX_NP_NORMAL_TILES)
            k_np_place_walk<EmitI16, true><<<wg2, 64, 0, ctx->stream>>>(dj, n_jobs, total_tiles, states, info, plan, res, done, c.res_mapped, tabs);
        else
            k_np_place_walk<EmitI16><<<wg2, 64, 0, ctx->stream>>>(dj, n_jobs, total_tiles, states, info, plan, res, done, c.res_mapped, tabs);
        VKX_LAUNCH_CHECK();
    }
    if (kind == VKX_NP_NORMAL_TILES) {
        VKX_TIMED(ctx, "k_np_tiles_finish");
        k_np_tiles_finish<<<dim3(vkx_blocks((size_t)c.max_tiles, 256), n_jobs), 256, 0, ctx->stream>>>(dj, info, plan);
        VKX_LAUNCH_CHECK();
    }
    if (!c.res_mapped)
        VKX_HIP(hipMemcpyAsync(c.results_host, res, (size_t)n_jobs * sizeof(vkx_np_result), hipMemcpyDeviceToHost, ctx->stream));
    return VKX_OK;
}

// Every job is one generator stream.  The jobs of one call are all of the normal family and of ONE kind, or all of the
// uniform family.  Asynchronous on the ctx stream: `results_host` (page-locked for a truly asynchronous copy) is valid
// after the stream has been synchronised.
// A large call runs as chunks of kChunkJobs jobs in software-pipelined form: the draw pass of a chunk (VALU bound) on the
// ctx stream, its resolve / place / walk passes (HBM and latency bound) on the context's second stream, so that the
// placement of chunk c shares the device with the draw of chunk c + 1; the tile arrays of consecutive chunks alternate
// between two scratch slots, and the ctx stream continues after the last chunk has been placed.
constexpr int kChunkJobs = 32;

int vkx_np_jobs_check(const vkx_np_job *jobs, int n_jobs, const vkx_np_result *results_host)
{
    VKX_REQUIRE(jobs && results_host, "NULL argument");
    VKX_REQUIRE(n_jobs >= 1 && n_jobs <= 65535, "1 .. 65535 jobs per call");
    const int kind = jobs[0].kind & 0xff;
    const bool uniform = kind == VKX_NP_CHOICE3_U8 || kind == VKX_NP_IMPULSE_U8;
    for (int i = 0; i < n_jobs; i++) {
        const vkx_np_job &j = jobs[i];
        VKX_REQUIRE(j.n >= 1 && j.n <= 0x7fffffffLL, "1 .. 2^31 - 1 samples per job");
        VKX_REQUIRE(j.dst != nullptr, "NULL destination");
        const int jkind = j.kind & 0xff;
        switch (jkind) {
        case VKX_NP_NORMAL_I16:
        case VKX_NP_NORMAL_TILES:
            VKX_REQUIRE(!uniform && jkind == kind, "the jobs of one call share a kind");
            VKX_REQUIRE(jkind != VKX_NP_NORMAL_TILES || ((uintptr_t)j.dst & 255) == 0, "tile buffers are 256-byte aligned");
            break;
        case VKX_NP_NORMAL_ADD_U8:
        case VKX_NP_SPECKLE_U8:
            VKX_REQUIRE(!uniform && jkind == kind, "the jobs of one call share a kind");
            VKX_REQUIRE(j.src != nullptr, "NULL source");
            break;
        case VKX_NP_CHOICE3_U8:
            VKX_REQUIRE(uniform, "normal and uniform jobs in one call");
            break;
        case VKX_NP_IMPULSE_U8:
            VKX_REQUIRE(uniform, "normal and uniform jobs in one call");
            VKX_REQUIRE(j.src != nullptr && j.cn >= 1 && j.cn <= 4, "bad impulse job");
            break;
        default:
            VKX_REQUIRE(false, "unknown job kind");
        }
        if (!uniform) VKX_REQUIRE(j.scale >= 0.0 && j.scale < 400.0, "scale outside [0, 400)");   // |z| < 40: int16 holds it
        VKX_REQUIRE(np_tiles_for(j, uniform) <= 262144, "stream too long for one job (2.6e8 samples)");
    }
    return VKX_OK;
}

// One chunk of checked jobs for callers that pipeline the passes themselves (vkx_chain_rgb_batch_np_dev, chain.hip).
struct vkx_np_chunk { NpChunk c; };

int vkx_np_chunk_begin(vkx_ctx *ctx, const vkx_np_job *jobs, int n_jobs, vkx_np_result *results_host, int slot, vkx_np_chunk **out)
{
    *out = nullptr;
    const NpTabs *tabs = nullptr;
    int rc = np_tables(ctx, &tabs);
    if (rc) return rc;
    vkx_np_chunk *h = new (std::nothrow) vkx_np_chunk();
    if (!h) { vkx_set_error("out of host memory"); return VKX_ERR_NOMEM; }
    NpChunk &c = h->c;
    c.jobs = jobs; c.n_jobs = n_jobs;
    c.kind = jobs[0].kind & 0xff;
    c.uniform = c.kind == VKX_NP_CHOICE3_U8 || c.kind == VKX_NP_IMPULSE_U8;
    c.slot = slot & 1;
    c.results_host = results_host;
    c.tabs = tabs;
    vkx_device_guard guard(ctx);
    if ((rc = np_chunk_prepare(ctx, c)) || (rc = np_chunk_front(ctx, c))) { delete h; return rc; }
    *out = h;
    return VKX_OK;
}

int vkx_np_chunk_finish(vkx_ctx *ctx, vkx_np_chunk *h)
{
    vkx_device_guard guard(ctx);
    return np_chunk_back(ctx, h->c);
}

void vkx_np_chunk_free(vkx_np_chunk *h) { delete h; }

VKX_EXPORT int vkx_np_draw_batch_dev(vkx_ctx *ctx, const vkx_np_job *jobs, int n_jobs, vkx_np_result *results_host)
{
    VKX_REQUIRE(ctx && jobs && results_host, "NULL argument");
    int rc = vkx_np_jobs_check(jobs, n_jobs, results_host);
    if (rc) return rc;
    const int kind = jobs[0].kind & 0xff;
    const bool uniform = kind == VKX_NP_CHOICE3_U8 || kind == VKX_NP_IMPULSE_U8;
    const NpTabs *tabs = nullptr;
    if ((rc = np_tables(ctx, &tabs))) return rc;
    vkx_device_guard guard(ctx);
    static const bool pipelined = [] { const char *e = getenv("VKX_NP_PIPELINE"); return !(e && e[0] == '0'); }();
    // (tile buffers: what follows the draw pass is a few microseconds per stream and no scratch grows with the records -- one chunk)
    const int n_chunks = (!uniform && pipelined && kind != VKX_NP_NORMAL_TILES && n_jobs >= 2 * kChunkJobs) ? (n_jobs + kChunkJobs - 1) / kChunkJobs : 1;
    auto chunk_of = [&](int k, NpChunk &c) {
        const int per = (n_jobs + n_chunks - 1) / n_chunks, first = k * per;
        c.jobs = jobs + first;
        c.n_jobs = std::min(per, n_jobs - first);
        c.kind = kind; c.uniform = uniform; c.slot = k & 1;
        c.results_host = results_host + first;
        c.tabs = tabs;
    };
    if (n_chunks == 1) {
        NpChunk c;
        chunk_of(0, c);
        if ((rc = np_chunk_prepare(ctx, c)) || (rc = np_chunk_front(ctx, c))) return rc;
        return np_chunk_back(ctx, c);
    }
    hipStream_t main_stream = ctx->stream;
    hipStream_t aux = vkx_stream_by_id(ctx, VKX_STREAM_COPY_IN, &rc);
    if (rc) return rc;
    std::vector<hipEvent_t> placed(n_chunks, nullptr);
    for (int k = 0; k < n_chunks; k++) {
        NpChunk c;
        chunk_of(k, c);
        // the slot's previous tenant (chunk k - 2) must have been placed before its arrays are overwritten
        if (k >= 2) VKX_HIP(hipStreamWaitEvent(main_stream, placed[k - 2], 0));
        if ((rc = np_chunk_prepare(ctx, c)) || (rc = np_chunk_front(ctx, c))) return rc;
        if ((rc = vkx_stream_order(ctx, aux, main_stream))) return rc;
        ctx->stream = aux;
        rc = np_chunk_back(ctx, c);
        ctx->stream = main_stream;
        if (rc) return rc;
        VKX_HIP(hipEventCreateWithFlags(&placed[k], hipEventDisableTiming));
        VKX_HIP(hipEventRecord(placed[k], aux));
    }
    for (int k = std::max(0, n_chunks - 2); k < n_chunks; k++) VKX_HIP(hipStreamWaitEvent(main_stream, placed[k], 0));
    for (hipEvent_t e : placed) (void)hipEventDestroy(e);      // destruction is deferred until the event has completed
    return VKX_OK;
}

// One job with HOST src / dst arrays: staged through the context, synchronous.
VKX_EXPORT int vkx_np_draw(vkx_ctx *ctx, const vkx_np_job *job, vkx_np_result *result_host)
{
    VKX_REQUIRE(ctx && job && result_host, "NULL argument");
    VKX_REQUIRE(job->n >= 1 && job->n <= 0x7fffffffLL, "1 .. 2^31 - 1 samples per job");
    VKX_REQUIRE(job->dst != nullptr, "NULL destination");
    size_t src_bytes = 0, dst_bytes = 0;
    switch (job->kind & 0xff) {
    case VKX_NP_NORMAL_I16: dst_bytes = (size_t)job->n * 2; break;
    case VKX_NP_NORMAL_TILES: dst_bytes = vkx_np_tiles_shape_of(job->n).bytes; break;     // the raw tile buffer (tests read it back)
    case VKX_NP_NORMAL_ADD_U8: case VKX_NP_SPECKLE_U8: src_bytes = dst_bytes = (size_t)job->n; break;
    case VKX_NP_CHOICE3_U8: dst_bytes = (size_t)job->n; break;
    case VKX_NP_IMPULSE_U8:
        VKX_REQUIRE(job->cn >= 1 && job->cn <= 4, "bad impulse job");
        src_bytes = dst_bytes = (size_t)job->n * job->cn;
        break;
    default: VKX_REQUIRE(false, "unknown job kind");
    }
    VKX_REQUIRE(src_bytes == 0 || job->src != nullptr, "NULL source");
    int rc = vkx_scratch_reserve(ctx, &ctx->stage[0], src_bytes ? src_bytes : 1);
    if (rc) return rc;
    if ((rc = vkx_scratch_reserve(ctx, &ctx->stage[1], dst_bytes))) return rc;
    vkx_np_job dev = *job;
    dev.src = src_bytes ? ctx->stage[0].ptr : nullptr;
    dev.dst = ctx->stage[1].ptr;
    {
        vkx_device_guard guard(ctx);
        if (src_bytes) VKX_HIP(hipMemcpyAsync(ctx->stage[0].ptr, job->src, src_bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    if ((rc = vkx_np_draw_batch_dev(ctx, &dev, 1, result_host))) return rc;
    vkx_device_guard guard(ctx);
    VKX_HIP(hipMemcpyAsync(job->dst, ctx->stage[1].ptr, dst_bytes, hipMemcpyDeviceToHost, ctx->stream));
    VKX_HIP(hipStreamSynchronize(ctx->stream));
    return VKX_OK;
}

// The int16 plane a finished VKX_NP_NORMAL_TILES buffer stands for (the staged chain, tests).  Asynchronous on the ctx stream.
VKX_EXPORT int vkx_np_tiles_expand_dev(vkx_ctx *ctx, const void *tiles, int64_t n, int16_t *dst)
{
    VKX_REQUIRE(ctx && tiles && dst, "NULL argument");
    VKX_REQUIRE(n >= 1 && n <= 0x7fffffffLL, "1 .. 2^31 - 1 samples");
    const vkx_np_tiles_shape s = vkx_np_tiles_shape_of(n);
    vkx_device_guard guard(ctx);
    VKX_TIMED(ctx, "k_np_tiles_expand");
    k_np_tiles_expand<<<vkx_blocks((size_t)s.n_tiles, 4), 256, 0, ctx->stream>>>((const uint2 *)((const unsigned char *)tiles + s.table_offset),
                                                                                 (const int16_t *)((const unsigned char *)tiles + s.slots_offset),
                                                                                 (int)s.n_tiles, n, dst);
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}
