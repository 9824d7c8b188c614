/* CPU restatement of the numpy.random.Generator streams the noise operators draw from -- TEST INFRASTRUCTURE ONLY
 * (same rules as vkx_oracle.c: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it).
 *
 * The reference's noise members call a numpy Generator (vkit/mechanism/distortion/photometric/noise.py:44-54
 * `rng.normal(0, std, shape)`, :100-157 `rng.choice((0, 1, 2), size, p)`, :160-190 `rng.normal`).  numpy is a
 * third-party dependency (setup.cfg: numpy >= 1.21) whose source is not under /root/reference; the algorithms below are
 * the published ones of numpy 1.17 ... 2.x:
 *   - PCG64 (numpy/random/src/pcg64/pcg64.h): 128-bit LCG, multiplier 0x2360ED051FC65DA44385DF649FCCF645, step THEN
 *     output XSL-RR 128/64 (rotr64(hi ^ lo, hi >> 58));
 *   - next_double = (next_uint64 >> 11) * 2^-53;
 *   - random_standard_normal (numpy/random/src/distributions/distributions.c): 256-layer ziggurat on one 64-bit draw
 *     (idx = r & 0xff, sign = bit 8, rabs = bits 9..60), wedge test with one more double, tail loop with pairs of
 *     doubles through log1p; random_normal = loc + scale * standard_normal;
 *   - Generator.choice(a, size, p) with replacement: cdf = cumsum(p) / cdf[-1], one double per element,
 *     searchsorted(side='right').
 * PINNED: tests/test_np_stream.py compares every function here with numpy itself (values and generator state), so this
 * file is a checked restatement, not a definition.  The tables come from the installed numpy (tools/np_tables.py). */
#include <math.h>
#include <stdint.h>
#include "../vkit_amd/csrc/np_ziggurat.h"

#define VKO_EXPORT __attribute__((visibility("default")))

typedef unsigned __int128 u128;

static const double kNorR = 3.6541528853610087963519472518;
static const double kNorInvR = 0.27366123732975827203338247596;

typedef struct { u128 state, inc; uint64_t draws; } vko_pcg;

static inline u128 pcg_mult(void) { return ((u128)0x2360ED051FC65DA4ull << 64) | 0x4385DF649FCCF645ull; }

static inline uint64_t pcg_next(vko_pcg *g)
{
    g->state = g->state * pcg_mult() + g->inc;
    g->draws++;
    const uint64_t hi = (uint64_t)(g->state >> 64), lo = (uint64_t)g->state;
    const uint64_t x = hi ^ lo;
    const unsigned r = (unsigned)(hi >> 58);
    return (x >> r) | (x << ((64 - r) & 63));
}
static inline double pcg_double(vko_pcg *g) { return (double)(pcg_next(g) >> 11) * (1.0 / 9007199254740992.0); }

static inline double bits2d(uint64_t b) { union { uint64_t u; double d; } v; v.u = b; return v.d; }

static double standard_normal(vko_pcg *g)
{
    for (;;) {
        uint64_t r = pcg_next(g);
        const int idx = (int)(r & 0xff);
        r >>= 8;
        const int sign = (int)(r & 1);
        const uint64_t rabs = (r >> 1) & 0x000fffffffffffffull;
        double x = (double)rabs * bits2d(kNpZigW[idx]);
        if (sign) x = -x;
        if (rabs < kNpZigK[idx]) return x;
        if (idx == 0) {
            for (;;) {
                const double xx = -kNorInvR * log1p(-pcg_double(g));
                const double yy = -log1p(-pcg_double(g));
                if (yy + yy > xx * xx) return ((rabs >> 8) & 1) ? -(kNorR + xx) : kNorR + xx;
            }
        } else {
            const double f1 = bits2d(kNpZigF[idx]), f0 = bits2d(kNpZigF[idx - 1]);
            if ((f0 - f1) * pcg_double(g) + f1 < exp(-0.5 * x * x)) return x;
        }
    }
}

static void load(vko_pcg *g, const uint64_t *st)
{
    g->state = ((u128)st[1] << 64) | st[0];
    g->inc = ((u128)st[3] << 64) | st[2];
    g->draws = 0;
}
static void store(const vko_pcg *g, uint64_t *st)
{
    st[0] = (uint64_t)g->state;
    st[1] = (uint64_t)(g->state >> 64);
}

/* st = {state lo, state hi, inc lo, inc hi}; the state words are advanced in place.  Returns the raw 64-bit draws used. */
VKO_EXPORT uint64_t vko_np_normal(uint64_t *st, int64_t n, double loc, double scale, double *out)
{
    vko_pcg g;
    load(&g, st);
    for (int64_t i = 0; i < n; i++) out[i] = loc + scale * standard_normal(&g);
    store(&g, st);
    return g.draws;
}

VKO_EXPORT uint64_t vko_np_random(uint64_t *st, int64_t n, double *out)
{
    vko_pcg g;
    load(&g, st);
    for (int64_t i = 0; i < n; i++) out[i] = pcg_double(&g);
    store(&g, st);
    return g.draws;
}

/* np.round(rng.normal(0, std, n)).astype(np.int16): rint = round half to even, C cast to int16 */
VKO_EXPORT uint64_t vko_np_normal_i16(uint64_t *st, int64_t n, double std, int16_t *out)
{
    vko_pcg g;
    load(&g, st);
    for (int64_t i = 0; i < n; i++) out[i] = (int16_t)(int32_t)rint(0.0 + std * standard_normal(&g));
    store(&g, st);
    return g.draws;
}

/* Generator.choice over len(p) = m categories: out[i] = #{k : cdf[k] <= u_i} */
VKO_EXPORT uint64_t vko_np_choice_cdf(uint64_t *st, int64_t n, const double *cdf, int m, uint8_t *out)
{
    vko_pcg g;
    load(&g, st);
    for (int64_t i = 0; i < n; i++) {
        const double u = pcg_double(&g);
        int k = 0;
        while (k < m && cdf[k] <= u) k++;
        out[i] = (uint8_t)k;
    }
    store(&g, st);
    return g.draws;
}

/* state after `delta` steps (pcg64 advance: O(log delta)) */
VKO_EXPORT void vko_np_advance(uint64_t *st, uint64_t delta_lo, uint64_t delta_hi)
{
    vko_pcg g;
    load(&g, st);
    u128 delta = ((u128)delta_hi << 64) | delta_lo;
    u128 acc_mult = 1, acc_plus = 0, cur_mult = pcg_mult(), cur_plus = g.inc;
    while (delta > 0) {
        if (delta & 1) {
            acc_mult *= cur_mult;
            acc_plus = acc_plus * cur_mult + cur_plus;
        }
        cur_plus = (cur_mult + 1) * cur_plus;
        cur_mult *= cur_mult;
        delta >>= 1;
    }
    g.state = acc_mult * g.state + acc_plus;
    store(&g, st);
}
