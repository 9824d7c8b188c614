#!/usr/bin/env python3
"""Generates tools/ubench/ubench.hip: one kernel per gfx950 instruction, 8 waves per SIMD on every CU, each wavefront
issuing a long unrolled run of that instruction on four independent destination registers.  The host part times every
kernel with HIP events and prints issue cycles per wavefront instruction per SIMD (4.0 = full rate for a 64-lane
wavefront on a 16-lane SIMD).  Used to price the instruction mix of k_chain_fused (profiles/r2_instruction_rates.md)."""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# name, asm with {d} (dst / accumulator), {a}, {b}, {c} sources, kind: '32' or '64' (register width of d/a/b/c)
T = [
    ('v_mad_u32_u24', 'v_mad_u32_u24 {d}, {a}, {b}, {d}', '32'),
    ('v_mul_u32_u24', 'v_mul_u32_u24 {d}, {a}, {d}', '32'),
    ('v_add_u32', 'v_add_u32 {d}, {a}, {d}', '32'),
    ('v_mul_lo_u32', 'v_mul_lo_u32 {d}, {a}, {d}', '32'),
    ('v_mul_hi_u32', 'v_mul_hi_u32 {d}, {a}, {d}', '32'),
    ('v_mad_u64_u32', 'v_mad_u64_u32 {d}, vcc, {a32}, {b32}, {d}', '64'),
    ('v_mad_i32_i24', 'v_mad_i32_i24 {d}, {a}, {b}, {d}', '32'),
    ('v_fma_f32', 'v_fma_f32 {d}, {a}, {b}, {d}', '32'),
    ('v_mul_f32', 'v_mul_f32 {d}, {a}, {d}', '32'),
    ('v_pk_fma_f32', 'v_pk_fma_f32 {d}, {a}, {b}, {d}', '64'),
    ('v_pk_mul_f32', 'v_pk_mul_f32 {d}, {a}, {d}', '64'),
    ('v_pk_add_f32', 'v_pk_add_f32 {d}, {a}, {d}', '64'),
    ('v_fma_f64', 'v_fma_f64 {d}, {a}, {b}, {d}', '64'),
    ('v_mul_f64', 'v_mul_f64 {d}, {a}, {d}', '64'),
    ('v_add_f64', 'v_add_f64 {d}, {a}, {d}', '64'),
    ('v_rcp_f64', 'v_rcp_f64 {d}, {d}', '64'),
    ('v_rcp_f32', 'v_rcp_f32 {d}, {d}', '32'),
    ('v_div_scale_f64', 'v_div_scale_f64 {d}, vcc, {a}, {b}, {a}', '64'),
    ('v_div_fmas_f64', 'v_div_fmas_f64 {d}, {a}, {b}, {d}', '64'),
    ('v_div_fixup_f64', 'v_div_fixup_f64 {d}, {a}, {b}, {d}', '64'),
    ('v_cvt_f32_f64', 'v_cvt_f32_f64 {d32}, {a}', '64'),
    ('v_cvt_f64_i32', 'v_cvt_f64_i32 {d}, {a32}', '64'),
    ('v_cvt_f64_f32', 'v_cvt_f64_f32 {d}, {a32}', '64'),
    ('v_cvt_i32_f64', 'v_cvt_i32_f64 {d32}, {a}', '64'),
    ('v_cvt_i32_f32', 'v_cvt_i32_f32 {d}, {a}', '32'),
    ('v_cvt_f32_i32', 'v_cvt_f32_i32 {d}, {a}', '32'),
    ('v_cvt_f32_ubyte0', 'v_cvt_f32_ubyte0 {d}, {a}', '32'),
    ('v_rndne_f32', 'v_rndne_f32 {d}, {d}', '32'),
    ('v_dot2_u32_u16', 'v_dot2_u32_u16 {d}, {a}, {b}, {d}', '32'),
    ('v_dot4_u32_u8', 'v_dot4_u32_u8 {d}, {a}, {b}, {d}', '32'),
    ('v_dot2_i32_i16', 'v_dot2_i32_i16 {d}, {a}, {b}, {d}', '32'),
    ('v_pk_mad_u16', 'v_pk_mad_u16 {d}, {a}, {b}, {d}', '32'),
    ('v_pk_add_u16', 'v_pk_add_u16 {d}, {a}, {d}', '32'),
    ('v_pk_mul_lo_u16', 'v_pk_mul_lo_u16 {d}, {a}, {d}', '32'),
    ('v_pk_max_i16', 'v_pk_max_i16 {d}, {a}, {d}', '32'),
    ('v_mad_u16', 'v_mad_u16 {d}, {a}, {b}, {d}', '32'),
    ('v_perm_b32', 'v_perm_b32 {d}, {a}, {d}, {b}', '32'),
    ('v_bfe_u32', 'v_bfe_u32 {d}, {d}, 8, 8', '32'),
    ('v_alignbit_b32', 'v_alignbit_b32 {d}, {a}, {d}, {b}', '32'),
    ('v_lshl_or_b32', 'v_lshl_or_b32 {d}, {d}, 8, {a}', '32'),
    ('v_and_or_b32', 'v_and_or_b32 {d}, {d}, {a}, {b}', '32'),
    ('v_add3_u32', 'v_add3_u32 {d}, {d}, {a}, {b}', '32'),
    ('v_med3_i32', 'v_med3_i32 {d}, {d}, {a}, {b}', '32'),
    ('v_max3_u32', 'v_max3_u32 {d}, {d}, {a}, {b}', '32'),
    ('v_sad_u32', 'v_sad_u32 {d}, {a}, {b}, {d}', '32'),
    ('v_lerp_u8', 'v_lerp_u8 {d}, {d}, {a}, {b}', '32'),
    ('v_cvt_pk_u8_f32', 'v_cvt_pk_u8_f32 {d}, {a}, {b}, {d}', '32'),
    ('v_ashr_pk_u8_i32', 'v_ashr_pk_u8_i32 {d}, {d}, {a}, {b}', '32'),
    ('v_bitop3_b32', 'v_bitop3_b32 {d}, {d}, {a}, {b} bitop3:0x96', '32'),
    ('v_cndmask_b32', 'v_cndmask_b32 {d}, {d}, {a}, vcc', '32'),
    ('v_cmp_lt_u32', 'v_cmp_lt_u32 vcc, {d}, {a}', '32'),
    ('v_lshlrev_b64', 'v_lshlrev_b64 {d}, 1, {d}', '64'),
    ('v_lshl_add_u64', 'v_lshl_add_u64 {d}, {d}, 1, {a}', '64'),
    ('v_mul_u32_u24_sdwa_byte', 'v_mul_u32_u24_sdwa {d}, {d}, {a} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD', '32'),
    ('v_add_u32_sdwa_word', 'v_add_u32_sdwa {d}, {d}, {a} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD', '32'),
    ('v_mov_b32_dpp_row_shr1', 'v_mov_b32_dpp {d}, {d} row_shr:1 row_mask:0xf bank_mask:0xf', '32'),
    ('v_mov_b32_dpp_wave_shr1', 'v_mov_b32_dpp {d}, {d} wave_shr:1 row_mask:0xf bank_mask:0xf', '32'),
    ('v_mul_u32_u24_dpp_wave_shr1', 'v_mul_u32_u24_dpp {d}, {d}, {a} wave_shr:1 row_mask:0xf bank_mask:0xf', '32'),
    ('v_add_u32_dpp_row_shr2', 'v_add_u32_dpp {d}, {d}, {a} row_shr:2 row_mask:0xf bank_mask:0xf', '32'),
    ('v_sub_u32', 'v_sub_u32 {d}, {a}, {d}', '32'),
    ('v_and_b32', 'v_and_b32 {d}, {a}, {d}', '32'),
    ('v_or_b32', 'v_or_b32 {d}, {a}, {d}', '32'),
    ('v_xor_b32', 'v_xor_b32 {d}, {a}, {d}', '32'),
    ('v_lshlrev_b32', 'v_lshlrev_b32 {d}, 3, {d}', '32'),
    ('v_lshrrev_b32', 'v_lshrrev_b32 {d}, 3, {d}', '32'),
    ('v_ashrrev_i32', 'v_ashrrev_i32 {d}, 3, {d}', '32'),
    ('v_lshlrev_b32_vshift', 'v_lshlrev_b32 {d}, {a}, {d}', '32'),
    ('v_max_u32', 'v_max_u32 {d}, {a}, {d}', '32'),
    ('v_min_i32', 'v_min_i32 {d}, {a}, {d}', '32'),
    ('v_mov_b32', 'v_mov_b32 {d}, {a}', '32'),
    ('v_add_f32', 'v_add_f32 {d}, {a}, {d}', '32'),
    ('v_sub_f32', 'v_sub_f32 {d}, {a}, {d}', '32'),
    ('v_max_f32', 'v_max_f32 {d}, {a}, {d}', '32'),
    ('v_fmac_f32', 'v_fmac_f32 {d}, {a}, {b}', '32'),
    ('v_mul_i32_i24', 'v_mul_i32_i24 {d}, {a}, {d}', '32'),
    ('v_lshl_add_u32', 'v_lshl_add_u32 {d}, {d}, 3, {a}', '32'),
    ('v_add_lshl_u32', 'v_add_lshl_u32 {d}, {d}, {a}, 3', '32'),
    ('v_xad_u32', 'v_xad_u32 {d}, {d}, {a}, {b}', '32'),
    ('v_or3_b32', 'v_or3_b32 {d}, {d}, {a}, {b}', '32'),
    ('v_cndmask_b32_e64_sgpr', 'v_cndmask_b32_e64 {d}, {d}, {a}, s[20:21]', '32'),
    ('v_cmp_lt_u32_e64_sgpr', 'v_cmp_lt_u32_e64 s[22:23], {d}, {a}', '32'),
    ('v_cmp_then_cndmask_pair', 'v_cmp_lt_u32_e32 vcc, {d}, {a}\\n\\tv_cndmask_b32_e32 {d}, {d}, {b}, vcc', '32'),
    ('v_add_co_u32', 'v_add_co_u32 {d}, vcc, {a}, {d}', '32'),
    ('v_readlane_b32', 'v_readlane_b32 s20, {d}, 3', '32'),
    ('v_readfirstlane_b32', 'v_readfirstlane_b32 s20, {d}', '32'),
    ('v_sub_u32_sdwa_byte', 'v_sub_u32_sdwa {d}, {d}, {a} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD', '32'),
    ('v_and_b32_dpp_row_shr1', 'v_and_b32_dpp {d}, {d}, {a} row_shr:1 row_mask:0xf bank_mask:0xf', '32'),
    ('v_fma_f32_neg', 'v_fma_f32 {d}, -{a}, {b}, {d}', '32'),
    ('v_mul_f32_dpp_row_shr1', 'v_mul_f32_dpp {d}, {d}, {a} row_shr:1 row_mask:0xf bank_mask:0xf', '32'),
    ('v_cvt_u32_f32', 'v_cvt_u32_f32 {d}, {d}', '32'),
    ('v_floor_f32', 'v_floor_f32 {d}, {d}', '32'),
    ('v_fract_f32', 'v_fract_f32 {d}, {d}', '32'),
    ('v_min3_u32', 'v_min3_u32 {d}, {d}, {a}, {b}', '32'),
    ('v_pk_add_i16', 'v_pk_add_i16 {d}, {a}, {d}', '32'),
    ('v_pk_min_i16', 'v_pk_min_i16 {d}, {a}, {d}', '32'),
    ('v_pk_lshrrev_b16', 'v_pk_lshrrev_b16 {d}, 3, {d}', '32'),
    ('v_fma_f32_x2_independent', 'v_fma_f32 {d}, {a}, {b}, {d}', '32'),
    ('v_permlane32_swap_b32', 'v_permlane32_swap_b32 {d}, {a}', '32'),
    ('ds_bpermute_b32', 'ds_bpermute_b32 {d}, {a}, {d}', '32lds'),
    ('ds_read_b32', 'ds_read_b32 {d}, {a}', '32lds'),
    ('ds_read_u16', 'ds_read_u16 {d}, {a}', '32lds'),
    ('ds_read_b64', 'ds_read_b64 {d}, {a32}', '64lds'),
    ('ds_read_b128_as4', 'ds_read_b128 {q}, {a32}', '128lds'),
    ('ds_write_b32', 'ds_write_b32 {a}, {d}', '32lds'),
    ('ds_max_u32', 'ds_max_u32 {a}, {d}', '32lds'),
]
UNROLL = 32


def kernel(idx, name, tmpl, kind):
    """Operands: %0-%3 64-bit accumulators d, %4 / %5 64-bit sources a / b, %6-%9 32-bit accumulators, %10 / %11 32-bit
    sources, %12 a 128-bit register (LDS reads).  A '32' kernel uses %6-%9 as d and %10 / %11 as a / b."""
    lds = kind.endswith('lds')
    base = kind.replace('lds', '')
    lines = []
    for u in range(UNROLL):
        k = u % 4
        if base == '32':
            s = tmpl.format(d=f'%{6 + k}', a='%10', b='%11')
        elif base == '64':
            s = tmpl.format(d=f'%{k}', a='%4', b='%5', d32=f'%{6 + k}', a32='%10', b32='%11')
        else:
            s = tmpl.format(q='%12', a32='%10')
        lines.append(s)
    if lds:
        lines.append('s_waitcnt lgkmcnt(0)')
    body = '\\n\\t'.join(lines)
    scale = {'32': 4, '64': 8, '128': 16}[base]
    clobber = (', "vcc"' if 'vcc' in tmpl else '') + (', "s20", "s21", "s22", "s23"' if 's2' in tmpl else '')
    return f'''
__global__ void __launch_bounds__(512, 8) k{idx}(uint32_t *out, int iters)
{{
    __shared__ uint32_t lds[4096];
    const uint32_t t = threadIdx.x, gid = blockIdx.x * 512 + t;
    lds[t] = t; lds[t + 512] = t;
    __syncthreads();
    double d0 = 1.0 + t * 1e-3, d1 = 1.5 + t * 1e-3, d2 = 2.0 + t * 1e-3, d3 = 2.5 + t * 1e-3, a = 1.0000001, b = 0.9999999;
    uint32_t e0 = t, e1 = t * 3u + 1u, e2 = t ^ 0x55u, e3 = t + 7u, a32 = (t & 63u) * {scale}u, b32 = 0x01020304u;
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    u4 q = {{0, 0, 0, 0}};
    for (int i = 0; i < iters; i++)
        asm volatile("{body}"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(a), "+v"(b), "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3),
                       "+v"(a32), "+v"(b32), "+v"(q) : : "memory"{clobber});
    out[gid] = (uint32_t)__double_as_longlong(d0 + d1 + d2 + d3) + e0 + e1 + e2 + e3 + q.x + q.y;
    if (iters < 0) out[0] = lds[t];
}}
'''


src = ['// generated by tools/ubench/gen.py -- do not edit', '#include <hip/hip_runtime.h>', '#include <stdio.h>',
       '#include <stdint.h>', '#include <vector>', '']
for i, (name, tmpl, kind) in enumerate(T):
    src.append(kernel(i, name, tmpl, kind))
src.append('typedef void (*kern_t)(uint32_t *, int);')
src.append('struct Entry { const char *name; kern_t fn; };')
src.append('static const Entry entries[] = {')
for i, (name, _, _) in enumerate(T):
    src.append(f'    {{"{name}", k{i}}},')
src.append('};')
src.append(f'''
int main()
{{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double mhz = prop.clockRate / 1000.0;      // kHz -> MHz
    const int blocks = cus * 4;                       // 4 x 512 threads = 32 wavefronts per CU = 8 per SIMD
    uint32_t *out;
    hipMalloc(&out, (size_t)blocks * 512 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000, unroll = {UNROLL};
    printf("device %s, %d CUs, %.0f MHz (reported)\\n", prop.name, cus, mhz);
    printf("| instruction | ns/launch | cycles per wavefront instruction per SIMD |\\n|---|---|---|\\n");
    for (const Entry &e : entries) {{
        e.fn<<<blocks, 512>>>(out, 10);
        hipDeviceSynchronize();
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {{
            hipEventRecord(e0);
            e.fn<<<blocks, 512>>>(out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }}
        // per SIMD: 8 wavefronts x iters x unroll instructions in `best` ms
        const double cycles = best * 1e-3 * mhz * 1e6 / (8.0 * iters * unroll);
        printf("| %s | %.0f | %.2f |\\n", e.name, best * 1e6, cycles);
    }}
    return 0;
}}
''')
with open(os.path.join(HERE, 'ubench.hip'), 'w') as f:
    f.write('\n'.join(src))
print('wrote', len(T), 'kernels')
