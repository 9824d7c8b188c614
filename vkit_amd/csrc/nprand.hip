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

constexpr int kRounds = 16;               // rounds of 64 draws per tile
constexpr int kTile = 64 * kRounds;       // raw draws per tile
constexpr double kNorR = 3.6541528853610087963519472518;
constexpr double kNorInvR = 0.27366123732975827203338247596;

#define PCG_MULT ((((u128)0x2360ED051FC65DA4ull) << 64) | (u128)0x4385DF649FCCF645ull)

// Stream-independent jump constants: s_{k + j} = A^j s_k + inc * G_j with G_j = 1 + A + ... + A^(j-1) (mod 2^128).
struct JumpTabs {
    uint64_t pow2[64][4];    // j = 2^i: A^j lo, hi, G_j lo, hi
    uint64_t lane[64][4];    // j = l + 1
    uint64_t a64[2], g64[2];
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
    long long n;             // samples wanted
    long long tile_base;     // index of the job's first tile in the batch-wide tile arrays
    int n_tiles;
    int kind, cn;
    double loc, scale;
    double cdf[3];
    const uint8_t *src;
    void *dst;
};

struct TileInfo {            // pass 1, assuming carry-in 0
    uint32_t count0, out0;
    uint64_t start0, emit0;  // round 0: positions that start an attempt / that emit a sample
};
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
__device__ __forceinline__ double u2dbl(uint64_t u) { return (double)(long long)(u >> 11) * (1.0 / 9007199254740992.0); }

// lane i <- lane i + 1 (lane 63 keeps `own`): DPP wave_shl:1
__device__ __forceinline__ uint32_t from_next_lane(uint32_t v, uint32_t own)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)own, (int)v, 0x130, 0xf, 0xf, false);
}

struct Attempt {
    double x;        // standard normal value when `emits`
    int len;         // raw draws the attempt consumes
    bool emits;
    bool inexact;    // x went through log1p: its last bits are not glibc's
};

// What the samples become.  pos = index of the sample in C order.
struct EmitNone {
    static constexpr bool kEnabled = false;
    __device__ void operator()(const NpJob &, long long, double, bool, uint32_t &) const {}
};
struct EmitI16 {   // np.round(0 + std * z).astype(int16)
    static constexpr bool kEnabled = true;
    __device__ void operator()(const NpJob &job, long long pos, double z, bool inexact, uint32_t &flags) const
    {
        const double v = job.scale * z;
        if (inexact) {
            const double f = v - floor(v);
            if (fabs(f - 0.5) < 1e-9) flags |= VKX_NP_AMBIGUOUS;
        }
        ((int16_t *)job.dst)[pos] = (int16_t)__double2int_rn(v);
    }
};
struct EmitAddU8 {   // clip(int16(px) + noise, 0, 255): the whole gaussion_noise operator
    static constexpr bool kEnabled = true;
    __device__ void operator()(const NpJob &job, long long pos, double z, bool inexact, uint32_t &flags) const
    {
        const double v = job.scale * z;
        if (inexact) {
            const double f = v - floor(v);
            if (fabs(f - 0.5) < 1e-9) flags |= VKX_NP_AMBIGUOUS;
        }
        const int k = (int16_t)__double2int_rn(v);
        const int s = (int16_t)((int)job.src[pos] + k);
        ((uint8_t *)job.dst)[pos] = (uint8_t)vkd::clamp_u8(s);
    }
};
struct EmitSpeckle {   // uint8(clip(px + px * (0 + std * z), 0, 255)) in float64
    static constexpr bool kEnabled = true;
    __device__ void operator()(const NpJob &job, long long pos, double z, bool inexact, uint32_t &flags) const
    {
        const double noise = 0.0 + job.scale * z;
        const double m = (double)job.src[pos];
        const double t = m * noise;
        double r = m + t;
        if (inexact && fabs(r - rint(r)) < 1e-9) flags |= VKX_NP_AMBIGUOUS;
        r = r < 0.0 ? 0.0 : (r > 255.0 ? 255.0 : r);
        ((uint8_t *)job.dst)[pos] = (uint8_t)(int)r;
    }
};

// The walk of one wavefront over one tile.  `base` = LCG state before the tile's first draw, `c_in` = leading draws
// already consumed by the previous tile's last attempt.  Lane l owns draws 64 r + l of the tile.
template <class Emit>
__device__ void walk_tile(const NpJob &job, const JumpTabs &g_jump, const uint4 *__restrict__ zig /* LDS: ki | wi */, const double *__restrict__ fi /* LDS */,
                          u128 base, uint32_t c_in, long long prefix, long long draw_base, const Emit &emit, uint32_t &count_out,
                          uint32_t &carry_out, uint64_t &start0, uint64_t &emit0, uint32_t &flags, unsigned long long *draws_used)
{
    const int lane = __lane_id();
    const u128 inc = mk128(job.inc);
    const u128 a64 = mk128(g_jump.a64), c64 = mk128(job.c64);
    // state after draw `lane` of the tile has been stepped
    u128 s = mk128(&g_jump.lane[lane][0]) * base + mk128(&g_jump.lane[lane][2]) * inc;
    uint32_t carry = c_in, count = 0;
    start0 = 0;
    emit0 = 0;
#pragma unroll 1
    for (int r = 0; r < kRounds; r++) {
        if (Emit::kEnabled && prefix + count >= job.n) break;   // everything wanted has been written
        const uint64_t u = pcg_out(s);
        const int idx = (int)(u & 0xff);
        const uint64_t rabs = (u >> 9) & 0x000fffffffffffffull;
        const uint4 e = zig[idx];
        const uint64_t ki = ((uint64_t)e.y << 32) | e.x;
        const double wi = __longlong_as_double(((long long)e.w << 32) | e.z);
        // rabs < 2^52: exact conversion through the exponent trick
        double x = (__longlong_as_double((long long)(0x4330000000000000ull | rabs)) - 4503599627370496.0) * wi;
        if (u & 0x100) x = -x;
        Attempt at;
        at.x = x; at.len = 1; at.emits = true; at.inexact = false;
        const bool fast = rabs < ki;
        const uint64_t slow = __ballot(!fast);
        if (slow) {
            // the draw after this one: the next lane's, or (lane 63) one own step
            uint64_t un;
            {
                const uint32_t lo = from_next_lane((uint32_t)u, 0u), hi = from_next_lane((uint32_t)(u >> 32), 0u);
                un = ((uint64_t)hi << 32) | lo;
            }
            if ((slow >> 63) && lane == 63) un = pcg_out(PCG_MULT * s + inc);
            if (!fast) {
                if (idx != 0) {
                    const double f1 = fi[idx], f0 = fi[idx - 1];
                    const double lhs = (f0 - f1) * u2dbl(un) + f1;
                    const double t = -0.5 * x;
                    const double rhs = exp(t * x);
                    at.len = 2;
                    at.emits = lhs < rhs;
                    if (fabs(rhs - lhs) <= lhs * 0x1p-42) flags |= VKX_NP_AMBIGUOUS;
                } else {
                    u128 t = s;
                    int len = 1;
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
                        if (l2 > r2) break;
                    }
                    at.len = len;
                    at.x = ((rabs >> 8) & 1) ? -(kNorR + xx) : kNorR + xx;
                    at.inexact = true;
                }
            }
        }
        // true starts of this round: positions not consumed by an earlier attempt
        uint64_t covered = carry >= 64 ? ~0ull : ((1ull << carry) - 1);
        carry = carry >= 64 ? carry - 64 : 0;
        uint64_t todo = slow & ~covered;
        while (todo) {
            const int j = __builtin_ctzll(todo);
            const int len = __builtin_amdgcn_readlane(at.len, j);
            const int end = j + len;                       // first position after the attempt
            uint64_t span;
            if (end >= 64) {
                span = j == 63 ? 0ull : (~0ull << (j + 1));
                if ((uint32_t)(end - 64) > carry) carry = (uint32_t)(end - 64);
            } else {
                span = ((1ull << end) - 1) & ~((2ull << j) - 1);
            }
            covered |= span;
            todo &= ~covered & ~(1ull << j);
        }
        const uint64_t starts = ~covered;
        const uint64_t emits = starts & __ballot(at.emits);
        if (r == 0) { start0 = starts; emit0 = emits; }
        if (Emit::kEnabled) {
            const bool mine = (emits >> lane) & 1;
            const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(emits >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)emits, 0u));
            const long long pos = prefix + count + rank;
            if (mine && pos < job.n) {
                emit(job, pos, at.x, at.inexact, flags);
                if (pos == job.n - 1) *draws_used = (unsigned long long)(draw_base + 64 * r + lane + at.len);
            }
        }
        count += (uint32_t)__builtin_popcountll(emits);
        s = a64 * s + c64;
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

__global__ void __launch_bounds__(256) k_np_tile_states(const NpJob *__restrict__ jobs, int n_jobs, long long total_tiles,
                                                        uint64_t *__restrict__ states /* [total_tiles][2] */, const NpTabs *__restrict__ tabs)
{
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

__global__ void __launch_bounds__(256) k_np_scan(const NpJob *__restrict__ jobs, int n_jobs, long long total_tiles,
                                                 const uint64_t *__restrict__ states, TileInfo *__restrict__ info,
                                                 vkx_np_result *__restrict__ results, const NpTabs *__restrict__ tabs)
{
    __shared__ uint4 zig[256];
    __shared__ double fi[256];
    load_tables(zig, fi, tabs);
    const JumpTabs &g_jump = tabs->jump;
    const long long n_waves = (long long)gridDim.x * 4;
    for (long long tile = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); tile < total_tiles; tile += n_waves) {
        const int j = job_of_tile(jobs, n_jobs, tile);
        const NpJob &job = jobs[j];
        uint32_t count, carry, flags = 0;
        uint64_t start0, emit0;
        walk_tile(job, g_jump, zig, fi, mk128(&states[2 * tile]), 0u, 0, 0, EmitNone(), count, carry, start0, emit0, flags, nullptr);
        if (__lane_id() == 0) {
            TileInfo t;
            t.count0 = count; t.out0 = carry; t.start0 = start0; t.emit0 = emit0;
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
            const uint32_t c = j ? info[j - 1].out0 : 0u;
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
                const uint32_t c = ((volatile TilePlan *)plan)[j].c_in;
                uint32_t count, out;
                if (c < 64 && ((t.start0 >> c) & 1)) {
                    count = t.count0 - (uint32_t)__builtin_popcountll(t.emit0 & ((1ull << c) - 1));
                    out = t.out0;
                } else {
                    uint64_t s0, e0;
                    walk_tile(job, g_jump, zig, fi, mk128(&st[2 * j]), c, 0, 0, EmitNone(), count, out, s0, e0, flags, nullptr);
                }
                if (threadIdx.x == 0) {
                    plan[j].count = count;
                    irregular[w] = m & ~(1ull << b);
                    if (j + 1 < T && out != t.out0) {
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
    for (int j = j0; j < j1; j++) {
        plan[j].prefix = run;
        run += plan[j].count;
    }
    if (threadIdx.x == 1023) {
        results[blockIdx.x].samples = part[1023];
        if ((long long)part[1023] < job.n) atomicOr(&results[blockIdx.x].flags, VKX_NP_SHORT);
    }
}

template <class Emit>
__global__ void __launch_bounds__(256) k_np_emit(const NpJob *__restrict__ jobs, int n_jobs, long long total_tiles,
                                                 const uint64_t *__restrict__ states, const TilePlan *__restrict__ plan,
                                                 vkx_np_result *__restrict__ results, const NpTabs *__restrict__ tabs)
{
    __shared__ uint4 zig[256];
    __shared__ double fi[256];
    load_tables(zig, fi, tabs);
    const JumpTabs &g_jump = tabs->jump;
    const long long n_waves = (long long)gridDim.x * 4;
    for (long long tile = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); tile < total_tiles; tile += n_waves) {
        const int j = job_of_tile(jobs, n_jobs, tile);
        const NpJob &job = jobs[j];
        const TilePlan p = plan[tile];
        if ((long long)p.prefix >= job.n) continue;
        uint32_t count, carry, flags = 0;
        uint64_t start0, emit0;
        walk_tile(job, g_jump, zig, fi, mk128(&states[2 * tile]), p.c_in, (long long)p.prefix, (tile - job.tile_base) * kTile, Emit(), count,
                  carry, start0, emit0, flags, &results[j].draws);
        if (flags) atomicOr(&results[j].flags, flags);
    }
}

// ---- uniform doubles: one draw per element (Generator.random / Generator.choice with p) -----------------------------
// impulse_noise: selector = #{k : cdf[k] <= u} per PIXEL (0 keep, 1 salt, 2 pepper), applied to all cn channels.
__global__ void __launch_bounds__(256) k_np_choice_impulse(const NpJob *__restrict__ jobs, int n_jobs, long long total_tiles,
                                                           const uint64_t *__restrict__ states, vkx_np_result *__restrict__ results,
                                                           const NpTabs *__restrict__ tabs)
{
    const JumpTabs &g_jump = tabs->jump;
    const long long n_waves = (long long)gridDim.x * 4;
    const int lane = __lane_id();
    for (long long tile = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); tile < total_tiles; tile += n_waves) {
        const int jb = job_of_tile(jobs, n_jobs, tile);
        const NpJob &job = jobs[jb];
        if (tile == job.tile_base && lane == 0) {
            results[jb].draws = (unsigned long long)job.n;
            results[jb].samples = (unsigned long long)job.n;
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

static long long np_tiles_for(const vkx_np_job &j, bool uniform)
{
    // raw draws to provision: the ziggurat uses 1.022 per sample on average
    const long long draws = uniform ? j.n : j.n + j.n / 32 + 2048;
    return (draws + kTile - 1) / kTile;
}

// Every job is one generator stream.  The jobs of one call are all of the normal family and of ONE kind, or all of the
// uniform family.  Asynchronous on the ctx stream: `results_host` (page-locked for a truly asynchronous copy) is valid
// after the stream has been synchronised.
VKX_EXPORT int vkx_np_draw_batch_dev(vkx_ctx *ctx, const vkx_np_job *jobs, int n_jobs, vkx_np_result *results_host)
{
    VKX_REQUIRE(ctx && jobs && results_host, "NULL argument");
    VKX_REQUIRE(n_jobs >= 1 && n_jobs <= 65535, "1 .. 65535 jobs per call");
    const int kind = jobs[0].kind;
    const bool uniform = kind == VKX_NP_CHOICE3_U8 || kind == VKX_NP_IMPULSE_U8;
    long long total_tiles = 0;
    int max_tiles = 0;
    for (int i = 0; i < n_jobs; i++) {
        const vkx_np_job &j = jobs[i];
        VKX_REQUIRE(j.n >= 1 && j.n <= 0x7fffffffLL, "1 .. 2^31 - 1 samples per job");
        VKX_REQUIRE(j.dst != nullptr, "NULL destination");
        switch (j.kind) {
        case VKX_NP_NORMAL_I16:
            VKX_REQUIRE(!uniform && j.kind == kind, "the jobs of one call share a kind");
            break;
        case VKX_NP_NORMAL_ADD_U8:
        case VKX_NP_SPECKLE_U8:
            VKX_REQUIRE(!uniform && j.kind == kind, "the jobs of one call share a kind");
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
        const long long tiles = np_tiles_for(j, uniform);
        VKX_REQUIRE(tiles <= 400000, "stream too long for one job");
        max_tiles = std::max<int>(max_tiles, (int)tiles);
        total_tiles += tiles;
    }
    const NpTabs *tabs = nullptr;
    int rc = np_tables(ctx, &tabs);
    if (rc) return rc;

    size_t off = 0;
    auto take = [&off](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_states = take((size_t)total_tiles * 16);
    const size_t o_info = take(uniform ? 0 : (size_t)total_tiles * sizeof(TileInfo));
    const size_t o_plan = take(uniform ? 0 : (size_t)total_tiles * sizeof(TilePlan));
    const size_t o_jobs = take((size_t)n_jobs * sizeof(NpJob));
    const size_t o_results = take((size_t)n_jobs * sizeof(vkx_np_result));
    rc = vkx_scratch_reserve(ctx, &ctx->np_work, off);
    if (rc) return rc;
    unsigned char *base = (unsigned char *)ctx->np_work.ptr;

    void *ring = nullptr;
    if ((rc = vkx_desc_ring_take(ctx, (size_t)n_jobs * sizeof(NpJob), &ring))) return rc;
    NpJob *hj = (NpJob *)ring;
    const u128 g64 = ((u128)g_jump_host.g64[1] << 64) | g_jump_host.g64[0];
    long long tile_base = 0;
    for (int i = 0; i < n_jobs; i++) {
        const vkx_np_job &j = jobs[i];
        NpJob &d = hj[i];
        d.state[0] = j.state[0]; d.state[1] = j.state[1];
        d.inc[0] = j.inc[0]; d.inc[1] = j.inc[1];
        const u128 c64 = (((u128)j.inc[1] << 64) | j.inc[0]) * g64;
        d.c64[0] = (uint64_t)c64; d.c64[1] = (uint64_t)(c64 >> 64);
        d.n = j.n;
        d.tile_base = tile_base;
        d.n_tiles = (int)np_tiles_for(j, uniform);
        tile_base += d.n_tiles;
        d.kind = j.kind; d.cn = j.cn;
        d.loc = 0.0; d.scale = j.scale;
        d.cdf[0] = j.cdf[0]; d.cdf[1] = j.cdf[1]; d.cdf[2] = j.cdf[2];
        d.src = (const uint8_t *)j.src;
        d.dst = j.dst;
    }
    vkx_device_guard guard(ctx);
    VKX_HIP(hipMemcpyAsync(base + o_jobs, hj, (size_t)n_jobs * sizeof(NpJob), hipMemcpyHostToDevice, ctx->stream));
    VKX_HIP(hipMemsetAsync(base + o_results, 0, (size_t)n_jobs * sizeof(vkx_np_result), ctx->stream));
    const NpJob *dj = (const NpJob *)(base + o_jobs);
    uint64_t *states = (uint64_t *)(base + o_states);
    TileInfo *info = (TileInfo *)(base + o_info);
    TilePlan *plan = (TilePlan *)(base + o_plan);
    vkx_np_result *res = (vkx_np_result *)(base + o_results);
    const unsigned wg = (unsigned)std::min<long long>((total_tiles + 3) / 4, 256 * 8);
    {
        VKX_TIMED(ctx, "k_np_tile_states");
        k_np_tile_states<<<vkx_blocks((size_t)total_tiles, 256), 256, 0, ctx->stream>>>(dj, n_jobs, total_tiles, states, tabs);
        VKX_LAUNCH_CHECK();
    }
    if (uniform) {
        VKX_TIMED(ctx, "k_np_choice_impulse");
        k_np_choice_impulse<<<wg, 256, 0, ctx->stream>>>(dj, n_jobs, total_tiles, states, res, tabs);
        VKX_LAUNCH_CHECK();
    } else {
        {
            VKX_TIMED(ctx, "k_np_scan");
            k_np_scan<<<wg, 256, 0, ctx->stream>>>(dj, n_jobs, total_tiles, states, info, res, tabs);
            VKX_LAUNCH_CHECK();
        }
        {
            VKX_TIMED(ctx, "k_np_resolve");
            const size_t bits = (((size_t)max_tiles + 63) / 64) * 8;
            k_np_resolve<<<n_jobs, 1024, bits, ctx->stream>>>(dj, states, info, plan, res, tabs);
            VKX_LAUNCH_CHECK();
        }
        {
            VKX_TIMED(ctx, "k_np_emit");
            if (kind == VKX_NP_NORMAL_I16)
                k_np_emit<EmitI16><<<wg, 256, 0, ctx->stream>>>(dj, n_jobs, total_tiles, states, plan, res, tabs);
            else if (kind == VKX_NP_NORMAL_ADD_U8)
                k_np_emit<EmitAddU8><<<wg, 256, 0, ctx->stream>>>(dj, n_jobs, total_tiles, states, plan, res, tabs);
            else
                k_np_emit<EmitSpeckle><<<wg, 256, 0, ctx->stream>>>(dj, n_jobs, total_tiles, states, plan, res, tabs);
            VKX_LAUNCH_CHECK();
        }
    }
    VKX_HIP(hipMemcpyAsync(results_host, res, (size_t)n_jobs * sizeof(vkx_np_result), hipMemcpyDeviceToHost, ctx->stream));
    return VKX_OK;
}

// One job with HOST src / dst arrays: staged through the context, synchronous.
VKX_EXPORT int vkx_np_draw(vkx_ctx *ctx, const vkx_np_job *job, vkx_np_result *result_host)
{
    VKX_REQUIRE(ctx && job && result_host, "NULL argument");
    VKX_REQUIRE(job->n >= 1 && job->n <= 0x7fffffffLL, "1 .. 2^31 - 1 samples per job");
    VKX_REQUIRE(job->dst != nullptr, "NULL destination");
    size_t src_bytes = 0, dst_bytes = 0;
    switch (job->kind) {
    case VKX_NP_NORMAL_I16: dst_bytes = (size_t)job->n * 2; break;
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
