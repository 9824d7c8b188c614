// numpy.random.Generator(PCG64).poisson(lam) with lam = the bytes of a uint8 image, drawn on the device from the caller's stream:
// vkit's poisson_noise (photometric/noise.py:81-90) is `rng.poisson(mat.astype(float32))` + a saturating narrow, and numpy (2.2.6,
// numpy/random/src/distributions/distributions.c: random_poisson / random_poisson_mult / random_poisson_ptrs / random_loggam, called
// per element in C order by _generator.pyx `disc` -> discrete_broadcast_d) takes a VARIABLE number of 64-bit draws per element:
//   lam == 0   none;
//   lam <  10  X + 1 draws (X the result): prod *= next_double until prod <= exp(-lam);
//   lam >= 10  two per attempt of the PTRS rejection loop (Hoermann's transformed rejection; acceptance 1 / invalpha = 75 .. 87 %).
// Where element i starts in the stream therefore depends on every draw before it -- the reason this member stayed on the host for
// three rounds (92 ms of sequential numpy per 1024^2 page).  The device path:
//   k_pz_stats   expected number of draws and its variance per block of 32 elements (both known in closed form per lam); the host turns
//                their prefix sums into SUPERBLOCKS of <= 256 blocks and, per block, the WINDOW of stream positions its first element
//                can start at: predicted position +- 6 sigma, relative to the exact start of the superblock;
//   k_pz_raw     the raw stream as doubles (next_double), positions 0 .. M;
//   k_pz_super   ONE launch of persistent workgroups (round 5; through round 4: one launch per superblock).  A workgroup takes the next
//                block in stream order: the outcome of EVERY (element, position) state of the block's band is evaluated once into LDS
//                (PTRS: accept / reject of the attempt that starts there -- the squeeze for all, the full test for the queued rest; lam < 10:
//                the draws the element takes from there), then one lane per candidate start walks the 32 elements through that table:
//                E[block][candidate] = where the next block starts.  The last block of a group of <= 16 walks all candidates of the group's
//                first window through the group's E rows (P: where a candidate stands at every row; G: where it enters the next group), the
//                last group of the superblock follows the exact start through the <= 16 G rows, publishes the next superblock's start and
//                reads every block's start out of P.  Windows relative to the start of the superblock `depth - 1` before keep the blocks of
//                `depth` superblocks in flight (see the host code);
//   k_pz_final   one lane per block walks its 32 elements from the exact start, now computing the values, and checks that it ends
//                where the next block begins.
// Every decision is numpy's: the per-lam constants, exp(-lam) and the loggam table are computed on the HOST with the same libm numpy
// calls (tests pin the table against numpy's own libnpyrandom.a); products, quotients, floor are IEEE double without contraction.  The
// two logarithms of the PTRS squeeze-free test are the device's (<= 1 ulp, like glibc's): a comparison closer than 2e-13 -- five times
// the worst-case sum of both error bounds -- raises VKX_NP_POISSON_AMBIGUOUS instead of guessing, as does a start that leaves its window (6 sigma), and
// the caller draws that image with numpy on the host.
#include "vkx_internal.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <mutex>
#include <vector>
#include <type_traits>

namespace {

typedef unsigned __int128 u128;
#define PZ_PCG_MULT ((((u128)0x2360ED051FC65DA4ull) << 64) | (u128)0x4385DF649FCCF645ull)

constexpr int kB = 32;              // elements per block
constexpr int kMaxBlocks = 256;     // blocks per superblock
constexpr int kBandMax = 3400;      // positions of a block's band (LDS: 8 bytes of draw + kB bytes of table each)
constexpr int kChainCap = kB * 3400 / 2 - 16 - 4096;    // E entries the chain stages at a time (the LDS of the tables)
constexpr int kKMax = 1024;         // loggam table: k + 1 <= kKMax
constexpr double kSigmas = 6.0;
constexpr int kCandThreads = 1024;
constexpr int kSlots = 8;           // rings of E / P / G rows: superblocks whose rows may exist at a time
constexpr int kGroupMax = 16;       // blocks per group of the two-level chain

enum { kFlagAmbiguous = 1, kFlagTable = 2, kFlagWindow = 4, kFlagMismatch = 8, kFlagDraws = 16, kFlagSize = 32 };

struct PzLam {                      // one per lam = 0 .. 255 (random_poisson_ptrs' locals; enlam of random_poisson_mult)
    double enlam, b, a, a2, vr, log_invalpha, loglam, lam;
};
struct PzTabs {
    PzLam lam[256];
    float mean[256], var[256];      // of the number of draws an element of that lam takes
    double loggam[kKMax + 1];       // random_loggam(x), x = 1 .. kKMax
};
struct PzBlock {
    int lo_rel;                     // first position of the window, relative to the exact start the superblock's windows refer to
    int W, band, e_off;             // candidates; positions the table covers; offset of the block's row in E (and of its row in P)
    int sup;                        // its superblock
    int g_off;                      // first block of a group: offset of the group's row in G
};
struct PzSuper {
    int first_block, n_blocks;
    int end_lo_rel, end_W;          // the window of the NEXT superblock's start
    int grp_rows, n_groups;         // blocks per group (the last group may hold fewer); groups
    int base;                       // the superblock whose exact start the windows are relative to (itself at depth 1)
    int grp0;                       // index of its first group counter
    int e_total, g_total;           // entries of its E rows; of its G rows
    int max_band, pad;
};

// ---- host: numpy's constants --------------------------------------------------------------------------------------
double np_loggam(double x)          // random_loggam, distributions.c (Zhang & Jin's asymptotic series + recurrence below 7)
{
    static const double a[10] = {8.333333333333333e-02, -2.777777777777778e-03, 7.936507936507937e-04, -5.952380952380952e-04,
                                 8.417508417508418e-04, -1.917526917526918e-03, 6.410256410256410e-03, -2.955065359477124e-02,
                                 1.796443723688307e-01, -1.39243221690590e+00};
    if (x == 1.0 || x == 2.0) return 0.0;
    long long n = x < 7.0 ? (long long)(7 - x) : 0;
    volatile double x0 = x + (double)n;
    const double x2 = (1.0 / x0) * (1.0 / x0);
    const double lg2pi = 1.8378770664093453e+00;
    double gl0 = a[9];
    for (int k = 8; k >= 0; k--) {
        gl0 *= x2;
        gl0 += a[k];
    }
    double gl = gl0 / x0 + 0.5 * lg2pi + (x0 - 0.5) * log(x0) - x0;
    if (x < 7.0) {
        for (long long k = 1; k <= n; k++) {
            gl -= log(x0 - 1.0);
            x0 = x0 - 1.0;
        }
    }
    return gl;
}

void build_tabs(PzTabs &T)
{
    memset(&T, 0, sizeof(T));
    for (int i = 0; i < 256; i++) {
        volatile double lam = (double)i;        // volatile: libm at run time, never the compiler's folding
        PzLam &e = T.lam[i];
        e.lam = lam;
        e.enlam = exp(-lam);
        if (i >= 10) {
            const double slam = sqrt(lam);
            e.loglam = log(lam);
            e.b = 0.931 + 2.53 * slam;
            e.a = -0.059 + 0.02483 * e.b;
            volatile double invalpha = 1.1239 + 1.1328 / (e.b - 3.4);
            e.vr = 0.9277 - 3.6224 / (e.b - 2);
            e.log_invalpha = log(invalpha);
            e.a2 = 2 * e.a;
            T.mean[i] = (float)(2.0 * invalpha);                        // attempts are geometric with success 1 / invalpha
            T.var[i] = (float)(4.0 * (invalpha * invalpha - invalpha));
        } else if (i > 0) {
            T.mean[i] = (float)(lam + 1.0);                             // X + 1 draws, X ~ Poisson(lam)
            T.var[i] = (float)lam;
        }
    }
    for (int x = 1; x <= kKMax; x++) T.loggam[x] = np_loggam((double)x);
}

void jump_consts(u128 j, u128 *a, u128 *g)      // s_{k + j} = A^j s_k + inc * G_j
{
    u128 acc_mult = 1, acc_plus = 0, cur_mult = PZ_PCG_MULT, cur_plus = 1;
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

// ---- device ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t pcg_out(u128 s)
{
    const uint64_t hi = (uint64_t)(s >> 64), lo = (uint64_t)s;
    const uint64_t x = hi ^ lo;
    const unsigned r = (unsigned)(hi >> 58);
    return (x >> r) | (x << ((64 - r) & 63));
}

// draws[i] = next_double of the stream's (i + 1)-th step.  Thread g owns positions g, g + T, g + 2 T, ... (T threads in the grid): one
// logarithmic jump to its first state, then the stride-T affine step.
__global__ void __launch_bounds__(256) k_pz_raw(uint64_t s_lo, uint64_t s_hi, uint64_t inc_lo, uint64_t inc_hi, uint64_t aT_lo, uint64_t aT_hi,
                                                uint64_t cT_lo, uint64_t cT_hi, long long M, double *__restrict__ draws)
{
    const long long T = (long long)gridDim.x * 256, g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= M) return;
    const u128 inc = ((u128)inc_hi << 64) | inc_lo, aT = ((u128)aT_hi << 64) | aT_lo, cT = ((u128)cT_hi << 64) | cT_lo;
    u128 s = ((u128)s_hi << 64) | s_lo;
    {
        u128 acc_mult = 1, acc_plus = 0, cur_mult = PZ_PCG_MULT, cur_plus = inc;
        for (unsigned long long j = (unsigned long long)g + 1; j > 0; j >>= 1) {
            if (j & 1) {
                acc_mult *= cur_mult;
                acc_plus = acc_plus * cur_mult + cur_plus;
            }
            cur_plus = (cur_mult + 1) * cur_plus;
            cur_mult *= cur_mult;
        }
        s = acc_mult * s + acc_plus;
    }
    for (long long i = g; i < M; i += T) {
        draws[i] = (double)(long long)(pcg_out(s) >> 11) * (1.0 / 9007199254740992.0);
        s = s * aT + cT;
    }
}

__global__ void __launch_bounds__(256) k_pz_stats(const uint8_t *__restrict__ src, long long n, const PzTabs *__restrict__ T,
                                                  float *__restrict__ bmean, float *__restrict__ bvar, float *__restrict__ brows)
{
    __shared__ float lm[256], lv[256];
    lm[threadIdx.x] = T->mean[threadIdx.x];
    lv[threadIdx.x] = T->var[threadIdx.x];
    __syncthreads();
    const long long blk = (long long)blockIdx.x * 256 + threadIdx.x, e0 = blk * kB;
    if (e0 >= n) return;
    float m = 0.f, v = 0.f;
    uint32_t seen[8] = {0, 0, 0, 0, 0, 0, 0, 0};          // the block's distinct values: the rows of its table (its cost; the plan deals heavy blocks first)
    auto mark = [&](int l) {
#pragma unroll
        for (int q = 0; q < 8; q++) seen[q] |= (l >> 5) == q ? 1u << (l & 31) : 0u;
    };
    if (e0 + kB <= n) {
        const uint4 *p = (const uint4 *)(src + e0);
#pragma unroll
        for (int q = 0; q < kB / 16; q++) {
            const uint4 w = p[q];
            const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int l = (ws[i >> 2] >> (8 * (i & 3))) & 0xff;
                m += lm[l];
                v += lv[l];
                mark(l);
            }
        }
    } else {
        for (long long e = e0; e < n; e++) {
            m += lm[src[e]];
            v += lv[src[e]];
            mark(src[e]);
        }
    }
    seen[0] &= ~1u;                                        // (zero takes no draws and has no row)
    int rows = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) rows += __popc(seen[q]);
    bmean[blk] = m;
    bvar[blk] = v;
    brows[blk] = (float)rows;
}

// random_loggam(x) for x >= 7 with the device's log: arguments beyond the host table (k + 1 > kKMax: us below ~1e-3 AND V <= us, a few
// attempts per million).  Differs from numpy's by the last bits of log(x) times x: the caller widens its ambiguity margin accordingly.
__device__ __forceinline__ double pz_loggam_dev(double x0)
{
    const double a[10] = {8.333333333333333e-02, -2.777777777777778e-03, 7.936507936507937e-04, -5.952380952380952e-04,
                          8.417508417508418e-04, -1.917526917526918e-03, 6.410256410256410e-03, -2.955065359477124e-02,
                          1.796443723688307e-01, -1.39243221690590e+00};
    const double x2 = (1.0 / x0) * (1.0 / x0);
    double gl0 = a[9];
#pragma unroll
    for (int k = 8; k >= 0; k--) {
        gl0 *= x2;
        gl0 += a[k];
    }
    return gl0 / x0 + 0.5 * 1.8378770664093453e+00 + (x0 - 0.5) * log(x0) - x0;
}

__device__ double pz_dbg[8];

// One attempt of random_poisson_ptrs' loop on the draws (d0, d1).  NEED_K: the value on the fast accept too.
// The comparison log(V) + log(invalpha) - log(a / us^2 + b) <= -lam + k log(lam) - loggam(k + 1) is first taken with float32 logarithms
// (v_log_f32; the left side is then within 3e-5 of the double one: |log V| <= 37, the other logarithm <= 74, both to 2e-7 relative):
// a difference beyond 1e-3 decides; the few per thousand inside it take the double logarithms, and those within 2e-13 of equality --
// where the device's log and glibc's could disagree -- raise the ambiguity flag.
template <bool NEED_K>
__device__ __forceinline__ bool pz_attempt(const PzLam &L, const double *__restrict__ loggam, double d0, double d1, double &kd, int &flags)
{
    const double U = d0 - 0.5, V = d1, us = 0.5 - fabs(U);
    const bool fast = us >= 0.07 && V <= L.vr;
    if (!NEED_K && fast) return true;
    kd = floor((L.a2 / us + L.b) * U + L.lam + 0.43);       // us == 0: -inf, rejected below like numpy's (int64)(-inf) < 0
    if (fast) return true;
    if (kd < 0.0 || (us < 0.013 && V > us)) return false;
    // the margin inside which the device's logarithms and glibc's could order the two sides differently: each library is within 1 ulp,
    // |log V| <= 37 and the other logarithm <= 74 (ulps of 7e-15 and 1.4e-14), so the two left sides differ by at most 4.2e-14
    constexpr double kTie = 2e-13;
    double lg, tol = kTie;
    if (kd < (double)kKMax) {
        lg = loggam[(int)kd + 1];
    } else {
        lg = pz_loggam_dev(kd + 1.0);
        tol = 1e-12 * kd * log(kd + 1.0) + 1e-9;
    }
    const double rhs = -L.lam + kd * L.loglam - lg;
    if (tol == kTie) {
        const float usf = (float)us;
        const float lhs_f = (__log2f((float)V) - __log2f((float)L.a * __builtin_amdgcn_rcpf(usf * usf) + (float)L.b)) * 0.69314718f;
        const double d = ((double)lhs_f + L.log_invalpha) - rhs;
        if (fabs(d) > 1e-3) return d < 0.0;
    }
    const double lhs = log(V) + L.log_invalpha - log(L.a / (us * us) + L.b);
    if (fabs(lhs - rhs) < tol) {
        flags |= kFlagAmbiguous;
        pz_dbg[0] = L.lam; pz_dbg[1] = d0; pz_dbg[2] = d1; pz_dbg[3] = lhs; pz_dbg[4] = rhs; pz_dbg[5] = kd;      // VKX_PZ_PROBE prints the last one
    }
    return lhs <= rhs;
}

constexpr uint8_t kCodeInvalid = 0xfe;
constexpr int kResolveRows = 12;         // table rows up to which the rejected PTRS states are resolved in the table (see k_pz_super)

// The attempt that starts at (d0, d1) as the table pass evaluates it: straight-line code (every lane takes the division, the float32
// logarithms and the loggam lookup, so that the compiler can overlap the states of consecutive elements), the double logarithms only for
// the lanes the float32 screen leaves open.  Same decisions as pz_attempt: it is the same arithmetic.
__device__ __forceinline__ bool pz_attempt_table(const PzLam &L, const double *__restrict__ lgam /* LDS */, double d0, double d1)
{
    const double U = d0 - 0.5, V = d1, us = 0.5 - fabs(U);
    const bool fast = us >= 0.07 && V <= L.vr;
    const double kd = floor((L.a2 / us + L.b) * U + L.lam + 0.43);
    const bool rej = kd < 0.0 || (us < 0.013 && V > us);
    const bool big = !(kd < (double)kKMax);
    const int ki = (int)fmin(fmax(kd, 0.0), (double)(kKMax - 1));
    const double rhs = -L.lam + kd * L.loglam - lgam[ki + 1];
    const float usf = (float)us;
    const float lhs_f = (__log2f((float)V) - __log2f((float)L.a * __builtin_amdgcn_rcpf(usf * usf) + (float)L.b)) * 0.69314718f;
    const double d = ((double)lhs_f + L.log_invalpha) - rhs;
    bool acc = fast || (!rej && d < 0.0);
    if (!fast && !rej && (big || !(fabs(d) > 1e-3))) {
        double kk;
        int flags = 0;
        acc = pz_attempt<false>(L, lgam, d0, d1, kk, flags);
    }
    return acc;
}

// All superblocks in ONE launch of persistent workgroups.  A workgroup takes the next block in stream order (a ticket), waits -- bounded --
// for the exact start its superblock's windows are relative to, and does the block: table, candidate walks, its E row.  The chain over
// a superblock's rows has two levels:
//   group   the workgroup that finishes last among the <= 16 blocks of a group stages the group's E rows in LDS (where its table was) and
//           walks EVERY candidate of the group's first window through them, one lane per candidate: P[row][c] = where candidate c stands at
//           that row, G[group][c] = where it enters the next group (16 dependent LDS reads, all candidates side by side);
//   top     the group that finishes last follows the superblock's exact start through the G rows (<= 16 dependent reads where the
//           one-level chain had 256), publishes the next superblock's start, and reads every block's start out of P in one round trip.
// A workgroup only ever waits for results of LOWER tickets, whose holders are running or done: no launch order, residency or grid size
// can deadlock it.  Between workgroups the rows travel as agent-scope stores and loads (write-through, past the per-XCD L2s; a
// __threadfence() here writes the L2 back, 25 - 40 us); E / P / G live in rings of kSlots superblocks, a slot is taken again when
// the superblock kSlots before has read its last row (`done`).
__device__ __forceinline__ long long pz_wait_ll(const long long *p, bool &timed_out)
{
    long long v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int spin = 0; v < 0 && spin < (1 << 22); spin++) {
        __builtin_amdgcn_s_sleep(8);
        v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (v < 0) timed_out = true;
    return v;
}

__global__ void __launch_bounds__(kCandThreads) k_pz_super(const uint8_t *__restrict__ src, long long n, int n_blk, const int *__restrict__ order, const PzBlock *__restrict__ blocks,
                                                           const PzSuper *__restrict__ supers, long long *pos /* exact starts of the superblocks; -1: not yet */,
                                                           const double *__restrict__ draws, long long M, const PzTabs *__restrict__ T,
                                                           uint16_t *E, uint16_t *P, uint16_t *G, long long e_stride /* entries per ring slot */,
                                                           long long *__restrict__ blk_pos, unsigned *ticket, unsigned *grp_counter, unsigned *sup_counter,
                                                           unsigned *done, int *__restrict__ fail, long long *__restrict__ probe, int g_rows_global, int resolve_rows)
{
#define PZ_STAMP(k) do { if (probe && tid == 0) probe[8 * sup + (k)] = (long long)wall_clock64(); } while (0)
    __shared__ double ld[kBandMax + 2];
    __shared__ __attribute__((aligned(16))) uint8_t tab[kB * (kBandMax + 1)];      // a row: band states + a sentinel
    __shared__ double lgam[kKMax + 1];
    __shared__ PzLam lL[kB];
    __shared__ int l_off[kMaxBlocks + 1], l_idx[kMaxBlocks];
    __shared__ uint8_t llam[kB], rowof[kB], rowlist[kB];
    __shared__ int s_last, s_idx, s_rows;
    __shared__ struct { double vr; int t, lam; } rinfo[kB];
    __shared__ uint32_t pq[kCandThreads / 64][128];                   // per wavefront: PTRS states past the squeeze, waiting for the full test
    __shared__ unsigned s_ticket;
    __shared__ long long s_p0;
    const int tid = threadIdx.x;
    for (int o = tid; o <= kKMax; o += kCandThreads) lgam[o] = T->loggam[o];
    for (;;) {
        __syncthreads();                                              // the LDS of the block before is free
        if (tid == 0) s_ticket = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned tk = (unsigned)__builtin_amdgcn_readfirstlane((int)s_ticket);
        if (tk >= (unsigned)n_blk) return;
        const int j = order[tk];            // (within a superblock the blocks are dealt by decreasing table size: see the host's plan)
        const PzBlock blk = blocks[j];
        const int sup = blk.sup;
        const PzSuper S = supers[sup];
        const int slot = sup % kSlots;
        uint16_t *Es = E + (long long)slot * e_stride, *Ps = P + (long long)slot * e_stride, *Gs = G + (long long)slot * e_stride;
        if (tid == 0) {
            bool timed_out = false;
            const long long ps = pz_wait_ll(pos + S.base, timed_out);
            if (sup >= kSlots) {
                const unsigned *dp = done + (sup - kSlots);
                unsigned d = __hip_atomic_load(dp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int spin = 0; !d && spin < (1 << 22); spin++) {
                    __builtin_amdgcn_s_sleep(8);
                    d = __hip_atomic_load(dp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (!d) timed_out = true;
            }
            if (timed_out) atomicOr(fail, kFlagWindow);
            s_p0 = timed_out ? 0 : ps;
        }
        __syncthreads();
        const long long p0 = s_p0;
        const bool last_of_super = j + 1 == S.first_block + S.n_blocks;
        if (last_of_super) PZ_STAMP(0);
        {
            const int next_lo_rel = last_of_super ? S.end_lo_rel : blocks[j + 1].lo_rel, next_W = last_of_super ? S.end_W : blocks[j + 1].W;
            const long long e0 = (long long)j * kB;
            const int n_el = (int)min((long long)kB, n - e0);
            const long long p_lo = p0 + blk.lo_rel;
            const int band = blk.band, stride = band + 1;
            if (tid < kB) llam[tid] = tid < n_el ? src[e0 + tid] : 0;
            if (tid < kB * 8) {                                       // the elements' constants: 8 doubles each
                const int t = tid >> 3;
                const int lam = t < n_el ? src[e0 + t] : 0;
                ((double *)lL)[tid] = ((const double *)&T->lam[lam])[tid & 7];
            }
            for (int o = tid; o < band + 2; o += kCandThreads) {
                const long long p = p_lo + o;
                ld[o] = p >= 0 && p < M ? draws[p] : -1.0;          // -1: no such draw
            }
            __syncthreads();
            if (tid < 64) {                                           // elements of one value share a table row
                int r = tid;
                const int mine = tid < kB ? llam[tid] : 0;
                for (int t = min(tid, kB) - 1; t >= 0; t--) r = llam[t] == mine ? t : r;
                if (tid < kB) rowof[tid] = (uint8_t)r;
                const bool is_row = tid < kB && r == tid && mine != 0;
                const unsigned long long rows_mask = __ballot(is_row);
                if (is_row) {
                    const int kk = __popcll(rows_mask & ((1ull << tid) - 1ull));
                    rowlist[kk] = (uint8_t)tid;
                    rinfo[kk].vr = lL[tid].vr;                        // what the table pass needs of a row, in one read
                    rinfo[kk].t = tid;
                    rinfo[kk].lam = mine;
                }
                if (tid == 0) s_rows = __popcll(rows_mask);
            }
            __syncthreads();
            if (last_of_super) PZ_STAMP(1);
            // the states of all rows as one index space, dealt round the lanes: every wavefront gets the same number of states whatever the
            // band's length (row by row, a band of 1 227 positions gave four of the sixteen wavefronts twice the others' work).  PTRS states:
            // the squeeze (us >= 0.07 and V <= vr: 80 - 87 % of the attempts accept right there) costs four double operations; the others
            // are queued per wavefront -- (row << 16 | position) words in LDS -- and evaluated 64 at a time with every lane busy: the
            // division, the two float32 logarithms and the loggam lookup run for the one state in seven that needs them.
            {
                const int total = s_rows * band, wave = tid >> 6, lane = tid & 63;
                uint32_t *q = pq[wave];
                int qn = 0;                                           // wave-uniform
                auto drain = [&](int first, int count) {
                    if (lane < count) {
                        const uint32_t w = q[first + lane];
                        const int t = (int)(w >> 16), o = (int)(w & 0xffffu);
                        const bool acc = pz_attempt_table(lL[t], lgam, ld[o], ld[o + 1]);
                        tab[t * stride + o] = acc ? 2 : 0;
                    }
                };
                int k = 0, o = tid;
                while (o >= band && k < s_rows) { o -= band; k++; }
                for (int base = 0; base < total; base += kCandThreads) {
                    const bool live = base + tid < total;
                    bool pend = false;
                    int t = 0;
                    if (live) {
                        const double vr = rinfo[k].vr;                // (one round trip: the row's record and the two draws together)
                        t = rinfo[k].t;
                        const int lam = rinfo[k].lam;
                        uint8_t code = kCodeInvalid;
                        if (lam >= 10) {
                            const double d0 = ld[o], d1 = ld[o + 1];
                            const double us = 0.5 - fabs(d0 - 0.5);
                            const bool fast = us >= 0.07 && d1 <= vr, valid = d0 >= 0.0 && d1 >= 0.0;
                            code = valid ? (fast ? 2 : 0) : kCodeInvalid;
                            pend = valid && !fast;
                        } else {
                            const double enlam = lL[t].enlam;
                            double prod = 1.0;
                            int c = 0;
                            while (o + c < band + 2 && c < 250) {
                                const double d = ld[o + c];
                                if (d < 0.0) break;
                                c++;
                                prod *= d;
                                if (!(prod > enlam)) {
                                    code = (uint8_t)c;
                                    break;
                                }
                            }
                        }
                        tab[t * stride + o] = code;
                    }
                    const unsigned long long pm = __ballot(pend);
                    if (pm) {
                        if (pend) q[qn + __popcll(pm & ((1ull << lane) - 1ull))] = ((uint32_t)t << 16) | (uint32_t)o;
                        qn += __popcll(pm);
                        if (qn >= 64) {
                            qn -= 64;
                            drain(qn, 64);
                        }
                    }
                    o += kCandThreads;
                    while (o >= band) { o -= band; k++; }
                }
                drain(0, qn);
            }
            if (tid < s_rows) tab[rowlist[tid] * stride + band] = kCodeInvalid;      // the sentinel a clamped position reads
            __syncthreads();
            // Few rows (a page: one or two values per block): every rejected PTRS state takes over the draws up to the accepted attempt of its
            // chain, in place -- a state read while its lane rewrites it is either still 0 (go on) or already its own total (add and stop) --,
            // and the walks below read ONE byte per element instead of one per attempt of the wavefront's unluckiest lane.  With many rows
            // (noise: 32 values per block) the pass costs more than the walks save.
            const bool resolved = s_rows <= resolve_rows;
            if (resolved) {
                for (int k = 0; k < s_rows; k++) {
                    const int t = rowlist[k];
                    if (llam[t] < 10) continue;
                    uint8_t *row = tab + t * stride;
                    for (int o = tid; o < band; o += kCandThreads) {
                        if (row[o] != 0) continue;
                        int oo = o, tot = 0;
                        uint32_t c = 0;
                        while (c == 0 && tot < 200) {
                            tot += 2;
                            oo += 2;
                            c = oo < band ? row[oo] : kCodeInvalid;
                        }
                        row[o] = c == 0 || c == kCodeInvalid || tot + (int)c > 250 ? kCodeInvalid : (uint8_t)(tot + (int)c);
                    }
                }
                __syncthreads();
            }
            if (last_of_super) PZ_STAMP(2);
            const int Wpad = (blk.W + 1) & ~1;                        // rows are stored as pairs (e_off is even)
            // Walks: lane = candidate start, NC per lane side by side (a window of 3 300 candidates costs the latency of one pass; the walk is
            // a chain of dependent LDS reads and the workgroup has four wavefronts per SIMD).  The elements' values and table rows sit in
            // wave-uniform registers.  RESOLVED (the table holds the draws up to the accepted attempt): one read per element, the position
            // clamped onto the row's sentinel instead of a range test.
            uint32_t lam_w[kB / 4], row_w[kB / 4];
#pragma unroll
            for (int q = 0; q < kB / 4; q++) {
                lam_w[q] = __builtin_amdgcn_readfirstlane(((const uint32_t *)llam)[q]);
                row_w[q] = __builtin_amdgcn_readfirstlane(((const uint32_t *)rowof)[q]);
            }
            auto walks = [&](auto nc_tag, auto res_tag) {
                constexpr int NC = decltype(nc_tag)::value;
                constexpr bool RESOLVED = decltype(res_tag)::value;
                int o[NC];
                bool ok[NC];
#pragma unroll
                for (int q = 0; q < NC; q++) {
                    const int c = tid + q * kCandThreads;
                    o[q] = c;
                    ok[q] = c < blk.W;
                }
#pragma unroll
                for (int t = 0; t < kB; t++) {
                    const int lam = (lam_w[t >> 2] >> (8 * (t & 3))) & 0xff;      // 0 beyond n_el
                    if (lam == 0) continue;
                    const uint8_t *row = tab + ((row_w[t >> 2] >> (8 * (t & 3))) & 0xff) * stride;
                    if (RESOLVED) {
                        uint32_t code[NC];
#pragma unroll
                        for (int q = 0; q < NC; q++) code[q] = row[min(o[q], band)];
#pragma unroll
                        for (int q = 0; q < NC; q++) {
                            ok[q] = ok[q] && code[q] != kCodeInvalid;
                            o[q] += (int)code[q];
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < NC; q++) {
                            uint8_t code = ok[q] && o[q] < band ? row[o[q]] : kCodeInvalid;
                            while (code == 0) {                  // a rejected PTRS attempt: the next one starts two draws on
                                o[q] += 2;
                                code = o[q] < band ? row[o[q]] : kCodeInvalid;
                            }
                            ok[q] = ok[q] && code != kCodeInvalid;
                            o[q] += code;
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < NC; q++) {
                    const int c = tid + q * kCandThreads;
                    const int e = o[q] + blk.lo_rel - next_lo_rel;
                    const uint32_t mine = ok[q] && e >= 0 && e < next_W ? (uint32_t)e : 0xffffu;
                    const uint32_t right = (uint32_t)__shfl_down((int)mine, 1);
                    if (!(c & 1) && c < Wpad)
                        __hip_atomic_store((uint32_t *)Es + ((blk.e_off + c) >> 1), mine | (right << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            };
            typedef std::integral_constant<int, 1> N1;
            typedef std::integral_constant<int, 2> N2;
            typedef std::integral_constant<int, 4> N4;
            if (resolved) {
                if (Wpad <= kCandThreads) walks(N1(), std::true_type());
                else if (Wpad <= 2 * kCandThreads) walks(N2(), std::true_type());
                else walks(N4(), std::true_type());
            } else {
                if (Wpad <= kCandThreads) walks(N1(), std::false_type());
                else if (Wpad <= 2 * kCandThreads) walks(N2(), std::false_type());
                else walks(N4(), std::false_type());
            }
        }
        // the last workgroup of the group to get here walks the group's rows
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (last_of_super) PZ_STAMP(3);
        const int grp = (j - S.first_block) / S.grp_rows, r0 = grp * S.grp_rows, cnt = min(S.grp_rows, S.n_blocks - r0);
        if (tid == 0) s_last = __hip_atomic_fetch_add(grp_counter + S.grp0 + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(cnt - 1);
        __syncthreads();
        if (!s_last) continue;
        uint16_t *le = (uint16_t *)tab;
        {
            const PzBlock fb = blocks[S.first_block + r0];
            const int base = fb.e_off & ~3;                                   // 8-byte loads: the staged rows start at `base`, row r at l_off[r] from there
            if (tid <= cnt) {
                const int r = r0 + tid;
                l_off[tid] = (r < S.n_blocks ? blocks[S.first_block + r].e_off : S.e_total) - base;
            }
            __syncthreads();
            const int n8 = (l_off[cnt] + 3) >> 2;
            const uint64_t *eg = (const uint64_t *)(Es + base);
            for (int i = tid; i < n8; i += kCandThreads) ((uint64_t *)le)[i] = __hip_atomic_load(eg + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const int W0 = fb.W, W0pad = (W0 + 1) & ~1;
            for (int c0 = 0; c0 < W0pad; c0 += kCandThreads) {
                const int c = c0 + tid;
                uint32_t cur = c < W0 ? (uint32_t)c : 0xffffu;
                for (int r = 0; r < cnt; r++) {
                    const int off = l_off[r];
                    const uint32_t right = (uint32_t)__shfl_down((int)cur, 1);
                    if (!(c & 1) && c < W0pad)
                        __hip_atomic_store((uint32_t *)Ps + ((base + off + c) >> 1), cur | (right << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t nx = le[off + (cur == 0xffffu ? 0u : cur)];
                    cur = cur == 0xffffu ? 0xffffu : nx;
                }
                const uint32_t right = (uint32_t)__shfl_down((int)cur, 1);
                if (!(c & 1) && c < W0pad)
                    __hip_atomic_store((uint32_t *)Gs + ((fb.g_off + c) >> 1), cur | (right << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // the last group of the superblock chains the G rows
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) s_last = __hip_atomic_fetch_add(sup_counter + sup, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(S.n_groups - 1);
        __syncthreads();
        if (!s_last) continue;
        PZ_STAMP(4);
        const bool g_in_lds = !g_rows_global && S.g_total <= kB * kBandMax / 2 - 8;      // (VKX_PZ_G_GLOBAL: the path of G rows beyond LDS, for tests)
        for (int g = tid; g < S.n_groups; g += kCandThreads) l_off[g] = blocks[S.first_block + g * S.grp_rows].g_off;
        if (g_in_lds) {
            const int n8 = (S.g_total + 3) >> 2;
            const uint64_t *gg = (const uint64_t *)Gs;
            for (int i = tid; i < n8; i += kCandThreads) ((uint64_t *)le)[i] = __hip_atomic_load(gg + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid == 0) {
            // This superblock's exact start: the base itself, or (windows relative to an earlier superblock, so that this one's tables ran
            // while the one before was still chaining) the value the superblock before publishes -- a bounded wait for a lower ticket.
            long long ps = p0;
            bool timed_out = false;
            if (S.base != sup) ps = pz_wait_ll(pos + sup, timed_out);
            const long long i0 = ps - (p0 + blocks[S.first_block].lo_rel);
            s_idx = !timed_out && ps >= 0 && i0 >= 0 && i0 < blocks[S.first_block].W ? (int)i0 : -1;
        }
        __syncthreads();
        if (tid < 64) {
            // wavefront 0 follows the start through the G rows: per step the row's offset (v_readlane), the read, a select
            uint32_t cur = s_idx < 0 ? 0xffffu : (uint32_t)s_idx;
            for (int gb = 0; gb < S.n_groups; gb += 64) {
                const int offv = l_off[min(gb + tid, S.n_groups - 1)];
                const int rows = min(64, S.n_groups - gb);
                for (int i = 0; i < rows; i++) {
                    const int off = __builtin_amdgcn_readlane(offv, i);
                    l_idx[gb + i] = (int)cur;
                    const uint32_t at = (uint32_t)off + (cur == 0xffffu ? 0u : cur);
                    uint32_t nx;
                    if (g_in_lds) {
                        nx = le[at];
                    } else {
                        const uint32_t w = __hip_atomic_load((const uint32_t *)Gs + (at >> 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        nx = (at & 1) ? w >> 16 : w & 0xffffu;
                    }
                    cur = cur == 0xffffu ? 0xffffu : nx;
                }
            }
            if (tid == 0) {
                s_idx = cur == 0xffffu ? -1 : (int)cur;
                if (s_idx < 0) atomicOr(fail, kFlagWindow);
                // (a failed superblock publishes a start all the same: its successors must not wait for it)
                __hip_atomic_store(pos + sup + 1, s_idx < 0 ? p0 : p0 + S.end_lo_rel + s_idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
        PZ_STAMP(5);
        // every block's exact start: one read of P at its group's start
        for (int b = tid; b < S.n_blocks; b += kCandThreads) {
            const uint32_t gi = (uint32_t)l_idx[b / S.grp_rows];
            long long v = -1;
            if (gi != 0xffffu) {
                const PzBlock bb = blocks[S.first_block + b];
                const uint32_t w = __hip_atomic_load((const uint32_t *)Ps + ((bb.e_off + (int)gi) >> 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t e = (gi & 1) ? w >> 16 : w & 0xffffu;
                if (e != 0xffffu) v = p0 + bb.lo_rel + (long long)e;
            }
            blk_pos[S.first_block + b] = v;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(done + sup, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        PZ_STAMP(6);
    }
}

#undef PZ_STAMP

__global__ void __launch_bounds__(64) k_pz_final(const uint8_t *__restrict__ src, long long n, long long n_blk, const long long *__restrict__ blk_pos,
                                                 const double *__restrict__ draws, long long M, const PzTabs *__restrict__ T,
                                                 uint8_t *__restrict__ dst, long long *__restrict__ consumed, int *__restrict__ fail)
{
    const long long b = (long long)blockIdx.x * 64 + threadIdx.x;
    if (b >= n_blk) return;
    long long p = blk_pos[b];
    const long long e0 = b * kB;
    const int n_el = (int)min((long long)kB, n - e0);
    int flags = 0;
    uint32_t out[kB / 4];
#pragma unroll
    for (int q = 0; q < kB / 4; q++) out[q] = 0;
    uint32_t in[kB / 4];
    if (n_el == kB) {
        const uint4 *sp = (const uint4 *)(src + e0);
#pragma unroll
        for (int q = 0; q < kB / 16; q++) {
            const uint4 w = sp[q];
            in[4 * q] = w.x; in[4 * q + 1] = w.y; in[4 * q + 2] = w.z; in[4 * q + 3] = w.w;
        }
    } else {
#pragma unroll
        for (int q = 0; q < kB / 4; q++) in[q] = 0;
        for (int t = 0; t < n_el; t++) in[t >> 2] |= (uint32_t)src[e0 + t] << (8 * (t & 3));
    }
    if (p < 0) flags |= kFlagDraws;
#pragma unroll
    for (int t = 0; t < kB; t++) {
        if (t >= n_el || flags) break;
        const int lam = (in[t >> 2] >> (8 * (t & 3))) & 0xff;
        if (lam == 0) continue;
        const PzLam &L = T->lam[lam];
        uint32_t val = 0;
        if (lam >= 10) {
            for (;;) {
                if (p + 2 > M) { flags |= kFlagDraws; break; }
                const double d0 = draws[p], d1 = draws[p + 1];
                p += 2;
                double kd = 0.0;
                if (pz_attempt<true>(L, T->loggam, d0, d1, kd, flags)) {
                    val = kd > 255.0 ? 255u : (kd < 0.0 ? 0u : (uint32_t)kd);
                    break;
                }
            }
        } else {
            const double enlam = L.enlam;
            double prod = 1.0;
            for (;;) {
                if (p + 1 > M) { flags |= kFlagDraws; break; }
                prod *= draws[p++];
                if (!(prod > enlam)) break;
                val++;
            }
        }
        out[t >> 2] |= min(val, 255u) << (8 * (t & 3));
    }
    if (n_el == kB) {
        uint4 *dp = (uint4 *)(dst + e0);
#pragma unroll
        for (int q = 0; q < kB / 16; q++) dp[q] = make_uint4(out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]);
    } else {
        for (int t = 0; t < n_el; t++) dst[e0 + t] = (uint8_t)(out[t >> 2] >> (8 * (t & 3)));
    }
    if (!flags) {
        if (b + 1 < n_blk) {
            if (p != blk_pos[b + 1]) flags |= kFlagMismatch;
        } else {
            *consumed = p;
        }
    }
    if (flags) atomicOr(fail, flags);
}

struct PzReply { long long consumed; int fail; int pad; };

}   // namespace

// out[i] = next_double of the PCG64 stream (state, inc) at its (i + 1)-th step, i = 0 .. M - 1; asynchronous on the ctx stream
int vkx_pcg64_doubles_dev(vkx_ctx *ctx, const uint64_t *state, const uint64_t *inc, long long M, double *out)
{
    if (M <= 0) return VKX_OK;
    const int grid = (int)std::min<long long>(1024, (M + 255) / 256);
    u128 aT, gT;
    jump_consts((u128)grid * 256, &aT, &gT);
    const u128 inc128 = ((u128)inc[1] << 64) | inc[0], cT = inc128 * gT;
    VKX_TIMED(ctx, "k_pz_raw");
    k_pz_raw<<<grid, 256, 0, ctx->stream>>>(state[0], state[1], inc[0], inc[1], (uint64_t)aT, (uint64_t)(aT >> 64), (uint64_t)cT, (uint64_t)(cT >> 64),
                                            M, out);
    VKX_LAUNCH_CHECK();
    return VKX_OK;
}

VKX_EXPORT int vkx_np_poisson_loggam_table(double *out, int n)
{
    if (!out || n < 1) return VKX_ERR_INVALID;
    for (int x = 1; x <= n; x++) out[x - 1] = np_loggam((double)x);
    return VKX_OK;
}

VKX_EXPORT int vkx_np_poisson_u8_dev(vkx_ctx *ctx, const uint64_t *state, const uint64_t *inc, const uint8_t *src, long long n, uint8_t *dst,
                          long long *consumed_host, unsigned *flags_host)
{
    if (!ctx || !state || !inc || !src || !dst || !consumed_host || !flags_host || n < 1 || n > (1ll << 30)) {
        vkx_set_error("vkx_np_poisson_u8: bad argument (1 <= n <= 2^30)");
        return VKX_ERR_INVALID;
    }
    vkx_device_guard guard(ctx);
    int rc;
    const long long n_blk = (n + kB - 1) / kB;
    // scratch: tables | block stats | block plan | positions | reply
    if (!ctx->pz_tabs_ready) {
        static PzTabs host_tabs;
        static std::once_flag host_once;              // contexts of several threads may get here together
        std::call_once(host_once, [] { build_tabs(host_tabs); });
        if ((rc = vkx_scratch_reserve(ctx, &ctx->pz_tabs, sizeof(PzTabs)))) return rc;
        VKX_HIP(hipMemcpyAsync(ctx->pz_tabs.ptr, &host_tabs, sizeof(PzTabs), hipMemcpyHostToDevice, ctx->stream));
        ctx->pz_tabs_ready = true;
    }
    const PzTabs *tabs = (const PzTabs *)ctx->pz_tabs.ptr;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_mean = 0, o_var = o_mean + up(sizeof(float) * n_blk), o_rows = o_var + up(sizeof(float) * n_blk), o_order = o_rows + up(sizeof(float) * n_blk),
                 o_plan = o_order + up(sizeof(int) * n_blk),
                 o_bpos = o_plan + up(sizeof(PzBlock) * n_blk), o_pos = o_bpos + up(sizeof(long long) * (n_blk + 1));
    // superblocks hold at least one block each: n_blk + 2 positions always suffice
    // counters: the ticket | per superblock: groups done, `done` flag | per group: blocks done
    const size_t o_counter = o_pos + up(sizeof(long long) * (n_blk + 2)), o_reply = o_counter + up(sizeof(unsigned) * (3 * (size_t)n_blk + 8)), work_bytes = o_reply + 256;
    if ((rc = vkx_scratch_reserve(ctx, &ctx->pz_work, work_bytes))) return rc;
    unsigned char *work = (unsigned char *)ctx->pz_work.ptr;
    float *d_mean = (float *)(work + o_mean), *d_var = (float *)(work + o_var), *d_rows = (float *)(work + o_rows);
    int *d_order = (int *)(work + o_order);
    PzBlock *d_plan = (PzBlock *)(work + o_plan);
    long long *d_bpos = (long long *)(work + o_bpos), *d_pos = (long long *)(work + o_pos);
    PzReply *d_reply = (PzReply *)(work + o_reply);
    unsigned *d_counter = (unsigned *)(work + o_counter);
    static const bool probing = getenv("VKX_PZ_PROBE") != nullptr;       // phase timestamps of every superblock (tools/poisson_probe.py)
    long long *d_probe = nullptr;

    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [&](const std::chrono::steady_clock::time_point &t0) { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); };
    double us_stats = 0, us_plan = 0, us_queue = 0;
    { VKX_TIMED(ctx, "k_pz_stats"); k_pz_stats<<<vkx_blocks((size_t)n_blk, 256), 256, 0, ctx->stream>>>(src, n, tabs, d_mean, d_var, d_rows); }
    VKX_LAUNCH_CHECK();
    // (host vectors of a call are kept per thread: a 1024^2 page is 98 k blocks, 4 MB of plan that would otherwise be mapped and faulted in
    //  anew by every call)
    static thread_local std::vector<float> bmean, bvar, brows;
    bmean.resize((size_t)n_blk); bvar.resize((size_t)n_blk); brows.resize((size_t)n_blk);
    VKX_HIP(hipMemcpyAsync(bmean.data(), d_mean, sizeof(float) * n_blk, hipMemcpyDeviceToHost, ctx->stream));
    VKX_HIP(hipMemcpyAsync(bvar.data(), d_var, sizeof(float) * n_blk, hipMemcpyDeviceToHost, ctx->stream));
    VKX_HIP(hipMemcpyAsync(brows.data(), d_rows, sizeof(float) * n_blk, hipMemcpyDeviceToHost, ctx->stream));
    VKX_HIP(hipStreamSynchronize(ctx->stream));
    us_stats = since(t_begin);

    // the plan: superblocks and windows (VKX_PZ_SIGMAS: narrower windows, to exercise the WINDOW refusal in tests)
    static const double sigmas = getenv("VKX_PZ_SIGMAS") ? atof(getenv("VKX_PZ_SIGMAS")) : kSigmas;
    // Depth 1: windows relative to the superblock's own start: its blocks begin when the superblock before has chained.  Depth d > 1:
    // relative to the start of the superblock d - 1 before, so that the tables and walks of d superblocks are in flight and a chain runs
    // under the tables of the next d - 1.  Wider windows are the price -- sqrt(d) more table states --, so dark images (the multiplication
    // method: a variance of lam per element against PTRS's 0.7 - 1.8) stay at depth 1.  VKX_PZ_DEPTH overrides.
    static const int depth_env = getenv("VKX_PZ_DEPTH") ? std::max(1, std::min(3, atoi(getenv("VKX_PZ_DEPTH")))) : 0;
    double var_sum = 0.0;
    for (long long b = 0; b < n_blk; b++) var_sum += bvar[(size_t)b];
    const double var_per_element = var_sum / (double)n;
    const int depth = depth_env ? depth_env : (var_per_element > 7.0 ? 1 : 2);
    static const int max_blocks_env = getenv("VKX_PZ_BLOCKS") ? std::max(16, std::min(kMaxBlocks, atoi(getenv("VKX_PZ_BLOCKS")))) : 0;
    const int max_blocks = max_blocks_env ? max_blocks_env : kMaxBlocks;
    std::vector<std::pair<double, double>> before;         // (mean, variance) of the depth - 1 superblocks before this one
    static thread_local std::vector<PzBlock> plan;
    static thread_local std::vector<PzSuper> supers;
    plan.resize((size_t)n_blk);
    supers.clear();
    double total_m = 0.0, total_v = 0.0, base_m = 0.0, base_v = 0.0;      // base: the superblock before (depth 2)
    long long e_max = 0;
    int n_groups_total = 0;
    for (long long s0 = 0; s0 < n_blk;) {
        PzSuper S;
        memset(&S, 0, sizeof(S));
        S.first_block = (int)s0;
        double cm = 0.0, cv = 0.0;
        long long e_off = 0;
        int max_band = 0, last_W = 1;
        long long j = s0;
        for (; j < n_blk && j - s0 < max_blocks; j++) {
            const bool exact = j == s0 && base_v == 0.0 && base_m == 0.0 && supers.empty();      // the stream's own start
            const int H = (j == s0 && depth == 1) || exact ? 0 : (int)ceil(sigmas * sqrt(base_v + cv)) + 3;
            const int W = 2 * H + 1;
            const int span = (int)ceil((double)bmean[j] + kSigmas * sqrt((double)bvar[j])) + 24;
            const int band = W + span;
            // (the variance cap keeps the first window of the NEXT superblock inside the band limit)
            if (j > s0 && (band > kBandMax || (depth > 1 && cv > 20000.0 / (depth - 1)))) break;
            PzBlock &b = plan[(size_t)j];
            b.lo_rel = (int)llround(base_m + cm) - H;
            b.W = W;
            b.band = std::min(band, kBandMax);
            b.e_off = (int)e_off;
            b.sup = (int)supers.size();
            b.g_off = 0;
            e_off += (W + 1) & ~1;          // rows start at even offsets: the entries are stored in pairs
            max_band = std::max(max_band, b.band);
            last_W = W;
            cm += bmean[j];
            cv += bvar[j];
        }
        S.n_blocks = (int)(j - s0);
        const int H = (int)ceil(sigmas * sqrt(base_v + cv)) + 3;
        S.end_lo_rel = (int)llround(base_m + cm) - H;
        S.end_W = 2 * H + 1;
        // groups of the two-level chain: as many rows as the LDS of a table holds (windows grow along the superblock: the last is the widest)
        S.grp_rows = std::max(1, std::min(kGroupMax, (kChainCap - 8) / (last_W + 1)));
        S.n_groups = (S.n_blocks + S.grp_rows - 1) / S.grp_rows;
        S.grp0 = n_groups_total;
        n_groups_total += S.n_groups;
        int g_off = 0;
        for (int g = 0; g < S.n_groups; g++) {
            PzBlock &fb = plan[(size_t)(s0 + (long long)g * S.grp_rows)];
            fb.g_off = g_off;
            g_off += (fb.W + 1) & ~1;
        }
        S.g_total = g_off;
        S.base = (int)supers.size() >= depth - 1 ? (int)supers.size() - (depth - 1) : 0;
        S.max_band = max_band;
        S.e_total = (int)e_off;
        e_max = std::max(e_max, e_off);
        total_m += cm;
        total_v += cv;
        if (depth > 1) {
            before.emplace_back(cm, cv);
            if ((int)before.size() > depth - 1) before.erase(before.begin());
            base_m = base_v = 0.0;
            for (const auto &q : before) { base_m += q.first; base_v += q.second; }
        }
        supers.push_back(S);
        s0 = j;
    }
    // Ticket order: superblock after superblock (a workgroup may only wait for lower tickets), and within a superblock the blocks with
    // the largest tables first -- a text row's block has up to 32 table rows, a white one a single row: dealt in index order the last heavy
    // block of a superblock starts when most light ones are done and the superblock's chain waits a whole table pass for it.
    static thread_local std::vector<int> order;
    order.resize((size_t)n_blk);
    static const bool heavy_first = !(getenv("VKX_PZ_ORDER") && atoi(getenv("VKX_PZ_ORDER")) == 0);
    // (a counting sort by table rows, more rows first; among equals the later block first: its window, hence its band, is the wider)
    for (const PzSuper &S : supers) {
        int *o = order.data() + S.first_block;
        if (!heavy_first) {
            for (int b = 0; b < S.n_blocks; b++) o[b] = S.first_block + b;
            continue;
        }
        int start[kB + 2];
        for (int r = 0; r <= kB + 1; r++) start[r] = 0;
        for (int b = 0; b < S.n_blocks; b++) start[kB - std::min(kB, (int)brows[(size_t)(S.first_block + b)]) + 1]++;
        for (int r = 1; r <= kB + 1; r++) start[r] += start[r - 1];
        for (int b = S.n_blocks - 1; b >= 0; b--) o[start[kB - std::min(kB, (int)brows[(size_t)(S.first_block + b)])]++] = S.first_block + b;
    }
    const long long M = (long long)ceil(total_m + (kSigmas + 1.0) * sqrt(total_v)) + 8192;
    if (M > (1ll << 29)) {          // more than 4 GB of raw draws: declined (the caller draws with numpy)
        *consumed_host = 0;
        *flags_host = kFlagSize;
        return VKX_OK;
    }
    us_plan = since(t_begin) - us_stats;
    const size_t e_slot_bytes = up(sizeof(uint16_t) * (size_t)e_max + 16), e_stride = e_slot_bytes / sizeof(uint16_t);
    const size_t draws_bytes = up(sizeof(double) * (size_t)(M + 2)), sup_bytes = up(sizeof(PzSuper) * supers.size());
    if ((rc = vkx_scratch_reserve(ctx, &ctx->pz_draws, draws_bytes + 3 * kSlots * e_slot_bytes + sup_bytes))) return rc;
    double *d_draws = (double *)ctx->pz_draws.ptr;
    uint16_t *d_E = (uint16_t *)((unsigned char *)ctx->pz_draws.ptr + draws_bytes);
    uint16_t *d_P = d_E + kSlots * e_stride, *d_G = d_P + kSlots * e_stride;
    PzSuper *d_sup = (PzSuper *)((unsigned char *)ctx->pz_draws.ptr + draws_bytes + 3 * kSlots * e_slot_bytes);
    if (probing) {
        if ((rc = vkx_scratch_reserve(ctx, &ctx->misc, 64 * supers.size() + 64))) return rc;
        d_probe = (long long *)ctx->misc.ptr;
        VKX_HIP(hipMemsetAsync(d_probe, 0, 64 * supers.size(), ctx->stream));
    }
    unsigned *d_ticket = d_counter, *d_supcnt = d_counter + 4, *d_done = d_supcnt + supers.size(), *d_grpcnt = d_done + supers.size();
    VKX_HIP(hipMemcpyAsync(d_plan, plan.data(), sizeof(PzBlock) * n_blk, hipMemcpyHostToDevice, ctx->stream));
    VKX_HIP(hipMemcpyAsync(d_order, order.data(), sizeof(int) * n_blk, hipMemcpyHostToDevice, ctx->stream));
    VKX_HIP(hipMemcpyAsync(d_sup, supers.data(), sizeof(PzSuper) * supers.size(), hipMemcpyHostToDevice, ctx->stream));
    VKX_HIP(hipMemsetAsync(d_pos, 0xff, sizeof(long long) * (supers.size() + 1), ctx->stream));      // -1: not published yet
    VKX_HIP(hipMemsetAsync(d_pos, 0, sizeof(long long), ctx->stream));
    VKX_HIP(hipMemsetAsync(d_reply, 0, sizeof(PzReply), ctx->stream));
    VKX_HIP(hipMemsetAsync(d_counter, 0, sizeof(unsigned) * (4 + 2 * supers.size() + (size_t)n_groups_total), ctx->stream));
    if ((rc = vkx_pcg64_doubles_dev(ctx, state, inc, M, d_draws))) return rc;
    {
        // persistent workgroups: one per CU (the tables take most of a CU's LDS); fewer blocks than CUs: one each
        static const int n_cu = [] {
            int dev = 0, v = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) v = 256;
            return v;
        }();
        static const int grid_env = getenv("VKX_PZ_GRID") ? atoi(getenv("VKX_PZ_GRID")) : 0;      // fewer workgroups than CUs: every wait of the kernel is exercised
        static const int g_global = getenv("VKX_PZ_G_GLOBAL") ? atoi(getenv("VKX_PZ_G_GLOBAL")) : 0;
        static const int resolve_rows = getenv("VKX_PZ_RESOLVE_ROWS") ? atoi(getenv("VKX_PZ_RESOLVE_ROWS")) : kResolveRows;
        // By default the kernel leaves one CU in eight alone (VKX_PZ_GRID=<CUs> takes them all).  Its workgroups are persistent, hold ~150
        // of a CU's 160 KB of LDS and spin on each other's published positions for 7 - 9 ms per page: with every CU taken, the kernels of the
        // OTHER worker processes sharing the GPU (the reference scales by processes) wait that long for a slot.  Measured with 8 workers on
        // C4 pages (one page in twenty-five draws poisson_noise): grid 256 -> 833 pages/s, 240 -> 807, 224 -> 1 044, 192 -> 1 088, 128 -> 993;
        // alone the page costs 7.4 ms at 256 and 8.1 at 224 (profiles/r6_poisson_grid_pool.txt).
        const int grid_default = std::max(1, n_cu - n_cu / 8);
        const int grid = (int)std::min<long long>(n_blk, grid_env > 0 ? grid_env : grid_default);
        VKX_TIMED(ctx, "k_pz_super");
        k_pz_super<<<grid, kCandThreads, 0, ctx->stream>>>(src, n, (int)n_blk, d_order, d_plan, d_sup, d_pos, d_draws, M, tabs, d_E, d_P, d_G, (long long)e_stride, d_bpos,
                                                          d_ticket, d_grpcnt, d_supcnt, d_done, &d_reply->fail, d_probe, g_global, resolve_rows);
    }
    VKX_LAUNCH_CHECK();
    { VKX_TIMED(ctx, "k_pz_final");
      k_pz_final<<<vkx_blocks((size_t)n_blk, 64), 64, 0, ctx->stream>>>(src, n, n_blk, d_bpos, d_draws, M, tabs, dst, &d_reply->consumed, &d_reply->fail); }
    VKX_LAUNCH_CHECK();
    us_queue = since(t_begin) - us_stats - us_plan;
    PzReply reply;
    VKX_HIP(hipMemcpyAsync(&reply, d_reply, sizeof(reply), hipMemcpyDeviceToHost, ctx->stream));
    VKX_HIP(hipStreamSynchronize(ctx->stream));
    if (probing)
        fprintf(stderr, "pz host (us): statistics kernel + read-back %.0f, plan %.0f, uploads + launches queued %.0f, wait for the device %.0f\n", us_stats, us_plan, us_queue,
                since(t_begin) - us_stats - us_plan - us_queue);
    if (probing && (reply.fail & kFlagAmbiguous)) {
        double dbg[8];
        VKX_HIP(hipMemcpyFromSymbol(dbg, HIP_SYMBOL(pz_dbg), sizeof(dbg)));
        fprintf(stderr, "pz ambiguous: lam %.0f d0 %.17g d1 %.17g lhs %.17g rhs %.17g k %.0f\n", dbg[0], dbg[1], dbg[2], dbg[3], dbg[4], dbg[5]);
    }
    if (probing) {
        std::vector<long long> pr(8 * supers.size());
        VKX_HIP(hipMemcpy(pr.data(), d_probe, 64 * supers.size(), hipMemcpyDeviceToHost));
        double acc[7] = {0, 0, 0, 0, 0, 0, 0};
        for (size_t q = 0; q < supers.size(); q++)
            for (int k = 1; k < 7; k++) acc[k] += (double)(pr[8 * q + k] - pr[8 * q + k - 1]);
        const double span = (double)(pr[8 * (supers.size() - 1) + 6] - pr[0]) / 100.0;
        fprintf(stderr, "pz probe (%zu superblocks, depth %d, mean us, 100 MHz clock): stage %.2f table %.2f walk %.2f groups %.2f chain %.2f fill %.2f; "
                        "first stamp to last %.1f us = %.2f per superblock; last superblock: blocks %d band %d W %d group rows %d\n",
                supers.size(), depth, acc[1] / supers.size() / 100, acc[2] / supers.size() / 100, acc[3] / supers.size() / 100, acc[4] / supers.size() / 100,
                acc[5] / supers.size() / 100, acc[6] / supers.size() / 100, span, span / supers.size(), supers.back().n_blocks, supers.back().max_band,
                supers.back().end_W, supers.back().grp_rows);
    }
    *consumed_host = reply.consumed;
    *flags_host = (unsigned)reply.fail;
    return VKX_OK;
}

VKX_EXPORT int vkx_np_poisson_u8(vkx_ctx *ctx, const uint64_t *state, const uint64_t *inc, const uint8_t *src_host, long long n, uint8_t *dst_host,
                      long long *consumed_host, unsigned *flags_host)
{
    if (!ctx || !src_host || !dst_host || n < 1 || n > (1ll << 30)) {
        vkx_set_error("vkx_np_poisson_u8: bad argument (1 <= n <= 2^30)");
        return VKX_ERR_INVALID;
    }
    vkx_device_guard guard(ctx);
    int rc;
    if ((rc = vkx_scratch_reserve(ctx, &ctx->stage[0], (size_t)n + 64))) return rc;
    if ((rc = vkx_scratch_reserve(ctx, &ctx->stage[1], (size_t)n + 64))) return rc;
    VKX_HIP(hipMemcpyAsync(ctx->stage[0].ptr, src_host, (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = vkx_np_poisson_u8_dev(ctx, state, inc, (const uint8_t *)ctx->stage[0].ptr, n, (uint8_t *)ctx->stage[1].ptr, consumed_host, flags_host))) return rc;
    VKX_HIP(hipMemcpyAsync(dst_host, ctx->stage[1].ptr, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    VKX_HIP(hipStreamSynchronize(ctx->stream));
    return VKX_OK;
}
