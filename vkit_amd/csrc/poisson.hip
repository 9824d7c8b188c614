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
//   k_pz_super   one launch per superblock, workgroup = block: the outcome of EVERY (element, position) state of the block's band is
//                evaluated once into LDS (PTRS: accept / reject of the attempt that starts there; lam < 10: the draws the element takes
//                from there), then one lane per candidate start walks the 32 elements through that table: E[block][candidate] = where
//                the next block starts; the workgroup that finishes last follows the exact start through the E rows (staged through
//                LDS): the exact start of every block and of the next superblock.  Launches of up to three superblocks are in flight
//                (windows relative to an earlier superblock's start; see the host code);
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
#include <mutex>
#include <vector>

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
    int lo_rel;                     // first position of the window, relative to the superblock's exact start
    int W, band, e_off;             // candidates; positions the table covers; offset of the block's row in E
};
struct PzSuper {
    int first_block, n_blocks;
    int end_lo_rel, end_W;          // the window of the NEXT superblock's start
    int chunk_blocks;               // rows the chain kernel stages at a time
    int max_band;
    long long e_total;
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
                                                  float *__restrict__ bmean, float *__restrict__ bvar)
{
    __shared__ float lm[256], lv[256];
    lm[threadIdx.x] = T->mean[threadIdx.x];
    lv[threadIdx.x] = T->var[threadIdx.x];
    __syncthreads();
    const long long blk = (long long)blockIdx.x * 256 + threadIdx.x, e0 = blk * kB;
    if (e0 >= n) return;
    float m = 0.f, v = 0.f;
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
            }
        }
    } else {
        for (long long e = e0; e < n; e++) {
            m += lm[src[e]];
            v += lv[src[e]];
        }
    }
    bmean[blk] = m;
    bvar[blk] = v;
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

// One superblock: workgroup = block (table, candidate walks, its E row); the workgroup that finishes last follows the exact start through
// the rows (staged through the LDS the tables occupied) and leaves the exact start of every block and of the next superblock.
// Between the workgroups of the launch the E rows travel as agent-scope stores and loads (write-through, past the per-XCD L2s): a
// __threadfence() here writes the L2 back, 25 - 40 us per workgroup.
__global__ void __launch_bounds__(kCandThreads) k_pz_super(const uint8_t *__restrict__ src, long long n, const PzBlock *__restrict__ blocks,
                                                           int first_block, int n_blocks, int end_lo_rel, int end_W, int chunk_blocks, int e_total,
                                                           const long long *pos_base /* the exact start the windows are relative to */,
                                                           long long *pos_cur /* this superblock's exact start (pos_base, or published by the superblock before) */,
                                                           const double *__restrict__ draws, long long M, const PzTabs *__restrict__ T,
                                                           uint16_t *E, long long *__restrict__ blk_pos, unsigned *counter,
                                                           int *__restrict__ fail, long long *__restrict__ probe)
{
#define PZ_STAMP(k) do { if (probe && tid == 0 && ((k) >= 4 || (int)blockIdx.x + 1 == n_blocks)) probe[k] = (long long)wall_clock64(); } while (0)
    __shared__ double ld[kBandMax + 2];
    __shared__ __attribute__((aligned(16))) uint8_t tab[kB * kBandMax];
    __shared__ double lgam[kKMax + 1];
    __shared__ PzLam lL[kB];
    __shared__ int l_off[kMaxBlocks + 1], l_lo[kMaxBlocks], l_idx[kMaxBlocks];
    __shared__ uint8_t llam[kB], rowof[kB], rowlist[kB];
    __shared__ int s_last, s_idx, s_rows;
    const int tid = threadIdx.x;
    const long long p0 = *pos_base;
    PZ_STAMP(0);
    {
        const int j = first_block + blockIdx.x;
        const PzBlock blk = blocks[j];
        const bool last = (int)blockIdx.x + 1 == n_blocks;
        const int next_lo_rel = last ? end_lo_rel : blocks[j + 1].lo_rel, next_W = last ? end_W : blocks[j + 1].W;
        const long long e0 = (long long)j * kB;
        const int n_el = (int)min((long long)kB, n - e0);
        const long long p_lo = p0 + blk.lo_rel;
        const int band = blk.band;
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
        for (int o = tid; o <= kKMax; o += kCandThreads) lgam[o] = T->loggam[o];
        __syncthreads();
        if (tid < 64) {                                           // elements of one value share a table row
            int r = tid;
            const int mine = tid < kB ? llam[tid] : 0;
            for (int t = min(tid, kB) - 1; t >= 0; t--) r = llam[t] == mine ? t : r;
            if (tid < kB) rowof[tid] = (uint8_t)r;
            const bool is_row = tid < kB && r == tid && mine != 0;
            const unsigned long long rows_mask = __ballot(is_row);
            if (is_row) rowlist[__popcll(rows_mask & ((1ull << tid) - 1ull))] = (uint8_t)tid;
            if (tid == 0) s_rows = __popcll(rows_mask);
        }
        __syncthreads();
        PZ_STAMP(1);
        // the states of all rows as one index space, dealt round the lanes: every wavefront gets the same number of states whatever the
        // band's length (row by row, a band of 1 227 positions gave four of the sixteen wavefronts twice the others' work)
        {
            const int total = s_rows * band;
            int k = 0, o = tid;
            while (o >= band && k < s_rows) { o -= band; k++; }
            for (int idx = tid; idx < total; idx += kCandThreads) {
                const int t = rowlist[k];
                const int lam = llam[t];
                const PzLam &L = lL[t];
                uint8_t code = kCodeInvalid;
                if (lam >= 10) {
                    const double d0 = ld[o], d1 = ld[o + 1];
                    const bool acc = pz_attempt_table(L, lgam, d0, d1);
                    code = d0 >= 0.0 && d1 >= 0.0 ? (acc ? 2 : 0) : kCodeInvalid;
                } else {
                    const double enlam = L.enlam;
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
                tab[t * band + o] = code;
                o += kCandThreads;
                while (o >= band) { o -= band; k++; }
            }
        }
        __syncthreads();
        PZ_STAMP(2);
        const int Wpad = (blk.W + 1) & ~1;                        // rows are stored as pairs (e_off is even)
        // the elements' values and table rows as wave-uniform words: the walk's only dependent LDS read is the table byte
        uint32_t lam_w[kB / 4], row_w[kB / 4];
#pragma unroll
        for (int q = 0; q < kB / 4; q++) {
            lam_w[q] = __builtin_amdgcn_readfirstlane(((const uint32_t *)llam)[q]);
            row_w[q] = __builtin_amdgcn_readfirstlane(((const uint32_t *)rowof)[q]);
        }
        for (int c0 = 0; c0 < Wpad; c0 += kCandThreads) {
            const int c = c0 + tid;
            int o = c;
            bool ok = c < blk.W;
#pragma unroll
            for (int t = 0; t < kB; t++) {
                const int lam = (lam_w[t >> 2] >> (8 * (t & 3))) & 0xff;      // 0 beyond n_el
                if (lam == 0) continue;
                const uint8_t *row = tab + ((row_w[t >> 2] >> (8 * (t & 3))) & 0xff) * band;
                uint8_t code = ok && o < band ? row[o] : kCodeInvalid;
                while (code == 0) {                  // a rejected PTRS attempt: the next one starts two draws on
                    o += 2;
                    code = o < band ? row[o] : kCodeInvalid;
                }
                ok = ok && code != kCodeInvalid;
                o += code;
            }
            const int e = o + blk.lo_rel - next_lo_rel;
            const uint32_t mine = ok && e >= 0 && e < next_W ? (uint32_t)e : 0xffffu;
            const uint32_t right = (uint32_t)__shfl_down((int)mine, 1);
            if (!(c & 1) && c < Wpad)
                __hip_atomic_store((uint32_t *)E + ((blk.e_off + c) >> 1), mine | (right << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // the last workgroup to get here chains
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    PZ_STAMP(3);
    if (tid == 0) s_last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(n_blocks - 1);
    __syncthreads();
    if (!s_last) return;
    PZ_STAMP(4);
    uint16_t *le = (uint16_t *)tab;
    for (int b = tid; b < n_blocks; b += kCandThreads) {
        l_off[b] = blocks[first_block + b].e_off;
        l_lo[b] = blocks[first_block + b].lo_rel;
    }
    if (tid == 0) {
        l_off[n_blocks] = e_total;
        // This superblock's exact start: the base itself, or (windows relative to the superblock BEFORE the previous one, so that this
        // launch's tables ran while the previous launch was still chaining) the value that launch publishes -- a bounded wait: launches
        // are queued in stream order, the publisher never waits for this one.
        long long ps = p0;
        if (pos_cur != pos_base) {
            ps = __hip_atomic_load(pos_cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int spin = 0; ps < 0 && spin < (1 << 22); spin++) {
                __builtin_amdgcn_s_sleep(16);
                ps = __hip_atomic_load(pos_cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        const long long i0 = ps - (p0 + blocks[first_block].lo_rel);
        s_idx = ps >= 0 && i0 >= 0 && i0 < blocks[first_block].W ? (int)i0 : -1;
    }
    __syncthreads();
    constexpr int kPer = (kChainCap * 2 / 8 + kCandThreads - 1) / kCandThreads + 1;      // 8-byte loads per thread and chunk
    uint64_t nextv[kPer];
    int next_n8 = 0;
    auto fetch = [&](int b0) {
        const int b1 = min(n_blocks, b0 + chunk_blocks);
        const int off0 = l_off[b0] & ~3, cnt = l_off[b1] - off0;
        const uint64_t *eg = (const uint64_t *)(E + off0);
        next_n8 = (cnt + 3) / 4;
#pragma unroll
        for (int q = 0; q < kPer; q++) {
            const int i = tid + q * kCandThreads;
            nextv[q] = i < next_n8 ? __hip_atomic_load(eg + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        }
    };
    fetch(0);
    for (int b0 = 0; b0 < n_blocks; b0 += chunk_blocks) {
        const int b1 = min(n_blocks, b0 + chunk_blocks);
        const int off0 = l_off[b0] & ~3;
#pragma unroll
        for (int q = 0; q < kPer; q++)
            if (tid + q * kCandThreads < next_n8) ((uint64_t *)le)[tid + q * kCandThreads] = nextv[q];
        __syncthreads();
        if (b1 < n_blocks) fetch(b1);                 // the next chunk's rows travel while lane 0 walks this one's
        if (tid < 64) {
            // Wavefront 0 walks the chunk: the dependent chain of a step is one add and one LDS read (every lane reads the same entry).
            // The rows' offsets sit in a vector register, one per lane, and come out with v_readlane; the starts found go back the same
            // way; a miss (0xffff) is noted on the side and the walk goes on with the index masked into the staged range.
            // Per step nothing but the offset (v_readlane), the masked add, the read and a fire-and-forget store of the index found: what
            // a miss (0xffff) means for the rows after it is sorted out when all chunks are done.
            uint32_t cur = s_idx < 0 ? 0xffffu : (uint32_t)s_idx;
            for (int bb = b0; bb < b1; bb += 64) {
                const int offv = l_off[min(bb + tid, n_blocks)] - off0;
                const int rows = min(64, b1 - bb);
                int *out = l_idx + bb;
                for (int i = 0; i < rows; i++) {
                    const int off = __builtin_amdgcn_readlane(offv, i);
                    out[i] = (int)cur;
                    cur = le[off + (cur & 0xfffu)];
                }
            }
            if (tid == 0) s_idx = cur == 0xffffu ? -1 : (int)cur;
        }
        __syncthreads();
    }
    // a row whose start is the miss marker, and every row after it, has no start
    if (tid < 64) {
        int first_bad = n_blocks;
        for (int b = tid; b < n_blocks; b += 64)
            if (l_idx[b] == 0xffff) first_bad = min(first_bad, b);
        for (int o = 32; o; o >>= 1) first_bad = min(first_bad, __shfl_xor(first_bad, o));
        if (tid == 0) {
            s_rows = first_bad;
            if (first_bad < n_blocks) s_idx = -1;
        }
    }
    __syncthreads();
    for (int b = tid; b < n_blocks; b += kCandThreads)
        if (b >= s_rows) l_idx[b] = -1;
    __syncthreads();
    PZ_STAMP(5);
    for (int b = tid; b < n_blocks; b += kCandThreads) blk_pos[first_block + b] = l_idx[b] < 0 ? -1 : p0 + l_lo[b] + l_idx[b];
    if (tid == 0) {
        if (s_idx < 0) atomicOr(fail, kFlagWindow);
        // (a failed superblock publishes a start all the same: its successors must not wait for it)
        __hip_atomic_store(pos_cur + 1, s_idx < 0 ? p0 : p0 + end_lo_rel + s_idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
    const size_t o_mean = 0, o_var = o_mean + up(sizeof(float) * n_blk), o_plan = o_var + up(sizeof(float) * n_blk),
                 o_bpos = o_plan + up(sizeof(PzBlock) * n_blk), o_pos = o_bpos + up(sizeof(long long) * (n_blk + 1));
    // superblocks hold at least one block each: n_blk + 2 positions always suffice
    const size_t o_counter = o_pos + up(sizeof(long long) * (n_blk + 2)), o_reply = o_counter + up(sizeof(unsigned) * (n_blk + 2)), work_bytes = o_reply + 256;
    if ((rc = vkx_scratch_reserve(ctx, &ctx->pz_work, work_bytes))) return rc;
    unsigned char *work = (unsigned char *)ctx->pz_work.ptr;
    float *d_mean = (float *)(work + o_mean), *d_var = (float *)(work + o_var);
    PzBlock *d_plan = (PzBlock *)(work + o_plan);
    long long *d_bpos = (long long *)(work + o_bpos), *d_pos = (long long *)(work + o_pos);
    PzReply *d_reply = (PzReply *)(work + o_reply);
    unsigned *d_counter = (unsigned *)(work + o_counter);
    static const bool probing = getenv("VKX_PZ_PROBE") != nullptr;       // phase timestamps of every superblock (tools/poisson_probe.py)
    long long *d_probe = nullptr;

    { VKX_TIMED(ctx, "k_pz_stats"); k_pz_stats<<<vkx_blocks((size_t)n_blk, 256), 256, 0, ctx->stream>>>(src, n, tabs, d_mean, d_var); }
    VKX_LAUNCH_CHECK();
    std::vector<float> bmean((size_t)n_blk), bvar((size_t)n_blk);
    VKX_HIP(hipMemcpyAsync(bmean.data(), d_mean, sizeof(float) * n_blk, hipMemcpyDeviceToHost, ctx->stream));
    VKX_HIP(hipMemcpyAsync(bvar.data(), d_var, sizeof(float) * n_blk, hipMemcpyDeviceToHost, ctx->stream));
    VKX_HIP(hipStreamSynchronize(ctx->stream));

    // the plan: superblocks and windows (VKX_PZ_SIGMAS: narrower windows, to exercise the WINDOW refusal in tests)
    static const double sigmas = getenv("VKX_PZ_SIGMAS") ? atof(getenv("VKX_PZ_SIGMAS")) : kSigmas;
    // Depth 1: windows relative to the superblock's own start, launches strictly one after the other.  Depth d > 1: relative to the start
    // of the superblock d - 1 before; launches go round d streams and the chain of one runs under the tables of the next d - 1 (the last
    // workgroup of a launch waits for the start its predecessor publishes).  Wider windows are the price -- sqrt(d) more table states --,
    // so dark images (the multiplication method: a variance of lam per element against PTRS's 0.7 - 1.8) stay at depth 1.  1024^2 RGB
    // page: 25.3 / 17.4 / 13.9 ms at depth 1 / 2 / 3; 512^2 x 3 of lam = 9: 9.0 / 9.5 / 11.5.  Default 2: with eight worker processes
    // sharing the GPU (tools/pool_scale.py) depth 3's extra table work and third stream cost more than its shorter critical path
    // gives (pipeline pages/s at 1 / 8 workers: depth 1 183 / 473, depth 2 192 / 540, depth 3 164 / 377).  VKX_PZ_DEPTH overrides.
    static const int depth_env = getenv("VKX_PZ_DEPTH") ? std::max(1, std::min(3, atoi(getenv("VKX_PZ_DEPTH")))) : 0;
    double var_sum = 0.0;
    for (long long b = 0; b < n_blk; b++) var_sum += bvar[(size_t)b];
    const double var_per_element = var_sum / (double)n;
    const int depth = depth_env ? depth_env : (var_per_element > 7.0 ? 1 : 2);
    const int max_blocks = kMaxBlocks - (depth - 1);      // CUs stay free for the workgroups still chaining
    std::vector<std::pair<double, double>> before;         // (mean, variance) of the depth - 1 superblocks before this one
    std::vector<PzBlock> plan((size_t)n_blk);
    std::vector<PzSuper> supers;
    double total_m = 0.0, total_v = 0.0, base_m = 0.0, base_v = 0.0;      // base: the superblock before (depth 2)
    long long e_max = 0;
    for (long long s0 = 0; s0 < n_blk;) {
        PzSuper S;
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
        S.chunk_blocks = std::max(1, (kChainCap - 8) / (last_W + 1));
        S.max_band = max_band;
        S.e_total = e_off;
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
    const long long M = (long long)ceil(total_m + (kSigmas + 1.0) * sqrt(total_v)) + 8192;
    if (M > (1ll << 29)) {          // more than 4 GB of raw draws: declined (the caller draws with numpy)
        *consumed_host = 0;
        *flags_host = kFlagSize;
        return VKX_OK;
    }
    if ((rc = vkx_scratch_reserve(ctx, &ctx->pz_draws, sizeof(double) * (size_t)(M + 2) + 3 * up(sizeof(uint16_t) * (size_t)e_max + 16)))) return rc;
    double *d_draws = (double *)ctx->pz_draws.ptr;
    uint16_t *d_E = (uint16_t *)((unsigned char *)ctx->pz_draws.ptr + up(sizeof(double) * (size_t)(M + 2)));
    if (probing) {
        if ((rc = vkx_scratch_reserve(ctx, &ctx->misc, 64 * supers.size() + 64))) return rc;
        d_probe = (long long *)ctx->misc.ptr;
        VKX_HIP(hipMemsetAsync(d_probe, 0, 64 * supers.size(), ctx->stream));
    }
    VKX_HIP(hipMemcpyAsync(d_plan, plan.data(), sizeof(PzBlock) * n_blk, hipMemcpyHostToDevice, ctx->stream));
    VKX_HIP(hipMemsetAsync(d_pos, 0xff, sizeof(long long) * (supers.size() + 1), ctx->stream));      // -1: not published yet
    VKX_HIP(hipMemsetAsync(d_pos, 0, sizeof(long long), ctx->stream));
    VKX_HIP(hipMemsetAsync(d_reply, 0, sizeof(PzReply), ctx->stream));
    VKX_HIP(hipMemsetAsync(d_counter, 0, sizeof(unsigned) * supers.size(), ctx->stream));
    if ((rc = vkx_pcg64_doubles_dev(ctx, state, inc, M, d_draws))) return rc;
    {
        hipStream_t main_stream = ctx->stream;
        hipStream_t lanes[3] = {main_stream, main_stream, main_stream};
        const int n_lanes = (int)std::min<size_t>((size_t)depth, supers.size());
        for (int k = 1; k < n_lanes; k++) {
            lanes[k] = vkx_stream_by_id(ctx, k == 1 ? VKX_STREAM_COPY_IN : VKX_STREAM_COPY_OUT, &rc);
            if (rc) return rc;
            if ((rc = vkx_stream_order(ctx, lanes[k], main_stream))) return rc;
        }
        const size_t e_stride = up(sizeof(uint16_t) * (size_t)e_max + 16) / sizeof(uint16_t);
        for (size_t s = 0; s < supers.size(); s++) {
            const PzSuper &S = supers[s];
            ctx->stream = lanes[s % (size_t)n_lanes];
            const size_t base = s >= (size_t)(depth - 1) ? s - (size_t)(depth - 1) : 0;
            {
                VKX_TIMED(ctx, "k_pz_super");
                k_pz_super<<<S.n_blocks, kCandThreads, 0, ctx->stream>>>(src, n, d_plan, S.first_block, S.n_blocks, S.end_lo_rel, S.end_W, S.chunk_blocks,
                                                                         (int)S.e_total, d_pos + base, d_pos + s, d_draws, M, tabs, d_E + (s % 3) * e_stride,
                                                                         d_bpos, d_counter + s, &d_reply->fail, d_probe ? d_probe + 8 * s : nullptr);
            }
            ctx->stream = main_stream;
            if (hipGetLastError() != hipSuccess) {
                vkx_set_error("k_pz_super: launch failed");
                return VKX_ERR_HIP;
            }
        }
        for (int k = 1; k < n_lanes; k++)
            if ((rc = vkx_stream_order(ctx, main_stream, lanes[k]))) return rc;
    }
    { VKX_TIMED(ctx, "k_pz_final");
      k_pz_final<<<vkx_blocks((size_t)n_blk, 64), 64, 0, ctx->stream>>>(src, n, n_blk, d_bpos, d_draws, M, tabs, dst, &d_reply->consumed, &d_reply->fail); }
    VKX_LAUNCH_CHECK();
    PzReply reply;
    VKX_HIP(hipMemcpyAsync(&reply, d_reply, sizeof(reply), hipMemcpyDeviceToHost, ctx->stream));
    VKX_HIP(hipStreamSynchronize(ctx->stream));
    if (probing && (reply.fail & kFlagAmbiguous)) {
        double dbg[8];
        VKX_HIP(hipMemcpyFromSymbol(dbg, HIP_SYMBOL(pz_dbg), sizeof(dbg)));
        fprintf(stderr, "pz ambiguous: lam %.0f d0 %.17g d1 %.17g lhs %.17g rhs %.17g k %.0f\n", dbg[0], dbg[1], dbg[2], dbg[3], dbg[4], dbg[5]);
    }
    if (probing) {
        std::vector<long long> pr(8 * supers.size());
        VKX_HIP(hipMemcpy(pr.data(), d_probe, 64 * supers.size(), hipMemcpyDeviceToHost));
        double acc[6] = {0, 0, 0, 0, 0, 0};
        for (size_t q = 0; q < supers.size(); q++)
            for (int k = 1; k < 6; k++) acc[k] += (double)(pr[8 * q + k] - pr[8 * q + k - 1]);
        fprintf(stderr, "pz probe (%zu superblocks, mean us, 100 MHz clock): stage %.2f table %.2f walk %.2f wait-for-last %.2f chain %.2f; last superblock: blocks %d band %d W %d chunk %d\n",
                supers.size(), acc[1] / supers.size() / 100, acc[2] / supers.size() / 100, acc[3] / supers.size() / 100, acc[4] / supers.size() / 100,
                acc[5] / supers.size() / 100, supers.back().n_blocks, supers.back().max_band, supers.back().end_W, supers.back().chunk_blocks);
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
