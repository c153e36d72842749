// tools/valu_rate.hip -- what one SIMD of gfx950 issues per cycle of the instructions the codec kernels live on.
//
//   hipcc --offload-arch=gfx950 -O2 tools/valu_rate.hip -o gpurun_out/valu_rate && gpurun_out/valu_rate > rates.jsonl
//
// Every test is one wave64 per workgroup running a loop of 64 independent instances of ONE instruction (eight
// destination registers in rotation), W = 1, 2, 4, 6, 8 waves per SIMD (4 W workgroups per CU, held there by a dynamic
// LDS allocation of 160 KiB / 4 W), grid = 256 CUs x 4 W, all resident at once.  Reported per test and W:
//   cyc_per_inst_simd   SIMD cycles per wave-instruction = a wave's s_memtime span / (instructions per wave x waves
//                       that shared its SIMD during that span, counted from HW_ID)
//   wall_cyc_per_inst   the same from the event-timed launch and the measured clock (cross-check)
//   clock_mhz           delta s_memtime / delta s_memrealtime x 100 MHz, median over waves
// "dep" variants chain one register (latency).  Mixed tests interleave a vector and a scalar instruction 1:1 to see
// whether the two pipes issue side by side.  Nothing here is product code; results go to profiles/r03_valu_rates.json.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned v4u __attribute__((ext_vector_type(4)));
struct Stamp { unsigned long long t0, t1, r0, r1; unsigned hwid, xcc; };

#define REP8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#define REP64(M) REP8(M) REP8(M) REP8(M) REP8(M) REP8(M) REP8(M) REP8(M) REP8(M)
#define REP32(M) REP8(M) REP8(M) REP8(M) REP8(M)

// One kernel per test.  The loop body is ONE asm statement of 64 instructions (the compiler pads every separate asm
// statement with s_nop against hazards it cannot see into; inside one statement the hazards are ours to respect).
// Operands: %0..%7 = x0..x7 (VGPR, read-write), %8..%15 = s0..s7 (SGPR), %16..%19 = mm0..mm3 (SGPR pairs),
//           %20 = a, %21 = b, %22 = c (VGPR inputs), %23 = m (SGPR pair), %24 = addr (VGPR, LDS byte address),
//           %25 = k (SGPR, a lane number)
#define X(J) "%" #J
#define XS(J) "%[s" #J "]"
#define XM(J) "%[m" #J "]"
#define KERNEL(NAME, ASMSTR)                                                                                     \
    __global__ void __launch_bounds__(64) k_##NAME(Stamp* st, unsigned* sink, int iters) {                       \
        extern __shared__ unsigned lds[];                                                                        \
        const unsigned lane = threadIdx.x;                                                                       \
        unsigned x0 = lane, x1 = lane * 3, x2 = lane * 5, x3 = lane * 7, x4 = lane * 11, x5 = lane * 13,       \
                 x6 = lane * 17, x7 = lane * 19;                                                                 \
        unsigned a = lane * 0x01010101u + 0x03020100u, b = 0x9e3779b1u * (lane + 1), c = 0x07060504u;          \
        unsigned s0 = sink[0], s1 = sink[1], s2 = sink[2], s3 = sink[3], s4 = sink[4], s5 = sink[5], s6 = sink[6], \
                 s7 = sink[7];                                                                                   \
        unsigned k = (sink[10] + 7) & 63;                                                                        \
        unsigned long long m = ((unsigned long long)sink[8] << 32) | sink[9] | 0x5555aaaa0f0ff0f0ull;            \
        unsigned long long mm0 = m, mm1 = m * 3, mm2 = m * 5, mm3 = m * 7;                                       \
        unsigned addr = (lane * 4) & 1023;                                                                       \
        unsigned long long y = lane;                                                                             \
        v4u z = {lane, lane, lane, lane};                                                            \
        const unsigned* gp = sink + 64;                                                                          \
        for (unsigned i = lane; i < 512; i += 64) lds[i] = (i * 0x01000193u) & 0x3fc;                            \
        __syncthreads();                                                                                         \
        unsigned long long t0, t1, r0, r1;                                                                       \
        unsigned hwid, xcc;                                                                                      \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));                                       \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                                       \
        asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0), "=s"(t0)::"memory"); \
        for (int it = 0; it < iters; ++it) {                                                                     \
            asm volatile(ASMSTR                                                                                  \
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7),       \
                           [s0] "+s"(s0), [s1] "+s"(s1), [s2] "+s"(s2), [s3] "+s"(s3), [s4] "+s"(s4),            \
                           [s5] "+s"(s5), [s6] "+s"(s6), [s7] "+s"(s7), [m0] "+s"(mm0), [m1] "+s"(mm1),          \
                           [m2] "+s"(mm2), [m3] "+s"(mm3), [y] "+v"(y), [z] "+v"(z)                              \
                         : [a] "v"(a), [b] "v"(b), [c] "v"(c), [m] "s"(m), [addr] "v"(addr), [k] "s"(k),         \
                           [gp] "s"(gp), [addr16] "v"((lane * 16) & 1023)                                        \
                         : "vcc", "scc", "memory");                                                              \
        }                                                                                                        \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" \
                     : "=s"(t1), "=s"(r1)::"memory");                                                            \
        unsigned acc = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7 ^ s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7 ^          \
                       (unsigned)(mm0 ^ mm1 ^ mm2 ^ mm3) ^ (unsigned)((mm0 ^ mm1 ^ mm2 ^ mm3) >> 32) ^           \
                       (unsigned)y ^ z.x ^ z.w;                                                                  \
        if (acc == 0x12345678u && iters < 0) sink[16 + lane] = acc;                                              \
        if (lane == 0) st[blockIdx.x] = Stamp{t0, t1, r0, r1, hwid, xcc};                                        \
    }

// ---- vector instructions, independent: destination x_J, eight in rotation ---------------------------------------
#define M_fma(J) "v_fma_f32 " X(J) ", %[a], %[b], " X(J) "\n\t"
#define M_add(J) "v_add_u32 " X(J) ", %[a], " X(J) "\n\t"
#define M_xor(J) "v_xor_b32 " X(J) ", %[a], " X(J) "\n\t"
#define M_lshl(J) "v_lshlrev_b32 " X(J) ", %[a], " X(J) "\n\t"
#define M_mov(J) "v_mov_b32 " X(J) ", %[a]\n\t"
#define M_perm(J) "v_perm_b32 " X(J) ", %[a], " X(J) ", %[c]\n\t"
#define M_movdpp_shr1(J) "v_mov_b32_dpp " X(J) ", %[a] row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define M_movdpp_bcast15(J) "v_mov_b32_dpp " X(J) ", %[a] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
#define M_movdpp_bcast31(J) "v_mov_b32_dpp " X(J) ", %[a] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
#define M_movdpp_waveshr(J) "v_mov_b32_dpp " X(J) ", %[a] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define M_ordpp_shr1(J) "v_or_b32_dpp " X(J) ", %[a], " X(J) " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define M_adddpp_shr1(J) "v_add_u32_dpp " X(J) ", %[a], " X(J) " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define M_cndmask_sgpr(J) "v_cndmask_b32_e64 " X(J) ", " X(J) ", %[a], %[m]\n\t"
#define M_cndmask_vcc(J) "v_cndmask_b32_e32 " X(J) ", " X(J) ", %[a], vcc\n\t"
#define M_alignbyte(J) "v_alignbyte_b32 " X(J) ", %[a], " X(J) ", %[b]\n\t"
#define M_alignbit(J) "v_alignbit_b32 " X(J) ", %[a], " X(J) ", %[b]\n\t"
#define M_and_or(J) "v_and_or_b32 " X(J) ", %[a], %[b], " X(J) "\n\t"
#define M_lshl_or(J) "v_lshl_or_b32 " X(J) ", %[a], %[b], " X(J) "\n\t"
#define M_lshl_add(J) "v_lshl_add_u32 " X(J) ", %[a], %[b], " X(J) "\n\t"
#define M_add3(J) "v_add3_u32 " X(J) ", %[a], %[b], " X(J) "\n\t"
#define M_bfi(J) "v_bfi_b32 " X(J) ", %[a], %[b], " X(J) "\n\t"
#define M_mul24_sdwa(J) "v_mul_u32_u24_sdwa " X(J) ", %[a], " X(J) " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t"
#define M_add_sdwa(J) "v_add_u32_sdwa " X(J) ", %[a], " X(J) " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\t"
#define M_mul24(J) "v_mul_u32_u24 " X(J) ", %[a], " X(J) "\n\t"
#define M_mad24(J) "v_mad_u32_u24 " X(J) ", %[a], %[b], " X(J) "\n\t"
#define M_mul_lo(J) "v_mul_lo_u32 " X(J) ", %[a], " X(J) "\n\t"
#define M_mul_hi(J) "v_mul_hi_u32 " X(J) ", %[a], " X(J) "\n\t"
#define M_bfe(J) "v_bfe_u32 " X(J) ", " X(J) ", %[c], 5\n\t"
#define M_ffbl(J) "v_ffbl_b32 " X(J) ", %[a]\n\t"
#define M_ffbh(J) "v_ffbh_u32 " X(J) ", %[a]\n\t"
#define M_bfrev(J) "v_bfrev_b32 " X(J) ", %[a]\n\t"
#define M_bcnt(J) "v_bcnt_u32_b32 " X(J) ", %[a], " X(J) "\n\t"
#define M_mbcnt(J) "v_mbcnt_lo_u32_b32 " X(J) ", %[k], " X(J) "\n\t"
#define M_sad_u8(J) "v_sad_u8 " X(J) ", %[a], %[b], " X(J) "\n\t"
#define M_dot4(J) "v_dot4_u32_u8 " X(J) ", %[a], %[b], " X(J) "\n\t"
#define M_cmp_sgpr(J) "v_cmp_lt_u32_e64 %[m" #J "], %[a], " X(J) "\n\t"
#define M_cmp_vcc(J) "v_cmp_lt_u32_e32 vcc, %[a], " X(J) "\n\t"
#define M_readlane(J) "v_readlane_b32 " XS(J) ", " X(J) ", 7\n\t"
#define M_readlane_sidx(J) "v_readlane_b32 " XS(J) ", " X(J) ", %[k]\n\t"
#define M_readfirstlane(J) "v_readfirstlane_b32 " XS(J) ", " X(J) "\n\t"
#define M_writelane(J) "v_writelane_b32 " X(J) ", %[k], 5\n\t"
#define M_swap(J) "v_swap_b32 " X(J) ", %[a]\n\t"
#define M_and(J) "v_and_b32 " X(J) ", %[a], " X(J) "\n\t"
#define M_or(J) "v_or_b32 " X(J) ", %[a], " X(J) "\n\t"
#define M_sub(J) "v_sub_u32 " X(J) ", %[a], " X(J) "\n\t"
#define M_subrev(J) "v_subrev_u32 " X(J) ", %[a], " X(J) "\n\t"
#define M_lshr(J) "v_lshrrev_b32 " X(J) ", %[a], " X(J) "\n\t"
#define M_lshl_const(J) "v_lshlrev_b32 " X(J) ", 3, " X(J) "\n\t"
#define M_min(J) "v_min_u32 " X(J) ", %[a], " X(J) "\n\t"
#define M_max(J) "v_max_u32 " X(J) ", %[a], " X(J) "\n\t"
#define M_not(J) "v_not_b32 " X(J) ", %[a]\n\t"
#define M_or3(J) "v_or3_b32 " X(J) ", %[a], %[b], " X(J) "\n\t"
#define M_xad(J) "v_xad_u32 " X(J) ", %[a], %[b], " X(J) "\n\t"
#define M_and_sgpr(J) "v_and_b32 " X(J) ", %[k], " X(J) "\n\t"
#define M_add_sgpr(J) "v_add_u32 " X(J) ", %[k], " X(J) "\n\t"
#define M_add_lit(J) "v_add_u32 " X(J) ", 0x12345, " X(J) "\n\t"
#define M_add_inline(J) "v_add_u32 " X(J) ", 7, " X(J) "\n\t"
#define M_add_e64(J) "v_add_u32_e64 " X(J) ", %[a], " X(J) "\n\t"
#define M_add_co(J) "v_add_co_u32 " X(J) ", vcc, %[a], " X(J) "\n\t"
#define M_addc_co(J) "v_addc_co_u32 " X(J) ", vcc, %[a], " X(J) ", vcc\n\t"
#define M_cndmask_e64_vcc(J) "v_cndmask_b32_e64 " X(J) ", " X(J) ", %[a], vcc\n\t"
#define M_cmp_cnd_vcc(J) "v_cmp_lt_u32_e32 vcc, %[a], " X(J) "\n\tv_cndmask_b32_e32 " X(J) ", " X(J) ", %[b], vcc\n\t"
#define M_cmp_cnd_sgpr(J) "v_cmp_lt_u32_e64 %[m0], %[a], " X(J) "\n\tv_cndmask_b32_e64 " X(J) ", " X(J) ", %[b], %[m0]\n\t"
#define M_pk_add_u16(J) "v_pk_add_u16 " X(J) ", %[a], " X(J) "\n\t"
#define M_add_u16(J) "v_add_u16 " X(J) ", %[a], " X(J) "\n\t"
#define M_and_sdwa(J) "v_and_b32_sdwa " X(J) ", %[a], " X(J) " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t"
#define M_mov_sdwa(J) "v_mov_b32_sdwa " X(J) ", %[a] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n\t"
#define M_fma_dep2(J) "v_fma_f32 " X(J) ", %[a], %[b], " X(J) "\n\t"
#define M_mul_f32(J) "v_mul_f32 " X(J) ", %[a], " X(J) "\n\t"
#define M_pk_fma(J) "v_pk_fma_f32 %[m" #J "], %[m" #J "], %[m" #J "], %[m" #J "]\n\t"
#define M_ds_read_b64(J) "ds_read_b64 %[y], %[addr16] offset:" #J "*64\n\t"
#define M_ds_read_b128(J) "ds_read_b128 %[z], %[addr16] offset:" #J "*64\n\t"
#define M_ds_write_b64(J) "ds_write_b64 %[addr16], %[y] offset:" #J "*64\n\t"
#define M_global_load(J) "global_load_dword " X(J) ", %[addr], %[gp] offset:" #J "*256\n\t"
#define M_branch_not_taken(J) "s_cbranch_scc0 Lend_%=\n\t"
#define M_branch_taken(J) "s_cbranch_scc1 Lt" #J "_%=\n\tLt" #J "_%=:\n\t"
#define REP8M4(M) M(0) M(1) M(2) M(3) M(0) M(1) M(2) M(3)
#define REP64M4(M) REP8M4(M) REP8M4(M) REP8M4(M) REP8M4(M) REP8M4(M) REP8M4(M) REP8M4(M) REP8M4(M)

// ---- LDS ------------------------------------------------------------------------------------------------------
#define M_ds_read_u8(J) "ds_read_u8 " X(J) ", %[addr] offset:" #J "*64\n\t"
#define M_ds_read_u16(J) "ds_read_u16 " X(J) ", %[addr] offset:" #J "*64\n\t"
#define M_ds_read_b32(J) "ds_read_b32 " X(J) ", %[addr] offset:" #J "*64\n\t"
#define M_ds_write_b8(J) "ds_write_b8 %[addr], " X(J) " offset:" #J "*64\n\t"
#define M_ds_write_b16(J) "ds_write_b16 %[addr], " X(J) " offset:" #J "*64\n\t"
#define M_ds_write_b32(J) "ds_write_b32 %[addr], " X(J) " offset:" #J "*64\n\t"
#define M_ds_or_b32(J) "ds_or_b32 %[addr], " X(J) " offset:" #J "*64\n\t"
#define M_ds_add_u32(J) "ds_add_u32 %[addr], " X(J) " offset:" #J "*64\n\t"
#define M_ds_bpermute(J) "ds_bpermute_b32 " X(J) ", %[addr], %[a]\n\t"
#define M_ds_permute(J) "ds_permute_b32 " X(J) ", %[addr], %[a]\n\t"
#define M_ds_swizzle(J) "ds_swizzle_b32 " X(J) ", %[a] offset:0x8000\n\t"
#define LDS_TAIL "s_waitcnt lgkmcnt(0)\n\t"

// ---- scalar ----------------------------------------------------------------------------------------------------
#define M_s_add(J) "s_add_u32 " XS(J) ", " XS(J) ", %[k]\n\t"
#define M_s_mov(J) "s_mov_b32 " XS(J) ", %[k]\n\t"
#define M_s_nop(J) "s_nop 0\n\t"
#define M_s_cselect(J) "s_cselect_b32 " XS(J) ", " XS(J) ", %[k]\n\t"
#define M_s_mul(J) "s_mul_i32 " XS(J) ", " XS(J) ", %[k]\n\t"
#define M_s_and64(J) "s_and_b64 " XM(J) ", " XM(J) ", %[m]\n\t"
#define M_s_andn2_64(J) "s_andn2_b64 " XM(J) ", %[m], " XM(J) "\n\t"
#define M_s_lshl64(J) "s_lshl_b64 " XM(J) ", " XM(J) ", 1\n\t"
#define M_s_bcnt64(J) "s_bcnt1_i32_b64 " XS(J) ", %[m]\n\t"
#define M_s_ff1_64(J) "s_ff1_i32_b64 " XS(J) ", %[m]\n\t"
#define M_s_bfm64(J) "s_bfm_b64 " XM(J) ", %[k], %[k]\n\t"
#define M_s_bitset0(J) "s_bitset0_b64 " XM(J) ", %[k]\n\t"

// ---- mixes: one vector + one other per slot, 32 + 32 per iteration ------------------------------------------------
#define M_mix_add_sadd(J) M_add(J) M_s_add(J)
#define M_mix_perm_sand(J) M_perm(J) "s_and_b32 " XS(J) ", " XS(J) ", %[k]\n\t"
#define M_mix_add_readlane(J) M_add(J) "v_readlane_b32 " XS(J) ", %[a], 7\n\t"
#define M_mix_add_dswrite(J) M_add(J) M_ds_write_b16(J)
#define M_mix_add_snop(J) M_add(J) "s_nop 0\n\t"
#define M_mix3(J) M_add(J) M_s_add(J) M_ds_write_b16(J)

// ---- dependent chains (latency): everything on x0 / s0 / mm0 -------------------------------------------------------
#define D_add(J) "v_add_u32 %0, %[a], %0\n\t"
#define D_perm(J) "v_perm_b32 %0, %[a], %0, %[c]\n\t"
#define D_movdpp(J) "s_nop 1\n\tv_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define D_ordpp(J) "s_nop 1\n\tv_or_b32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define D_mul_lo(J) "v_mul_lo_u32 %0, %[a], %0\n\t"
#define D_ds_read(J) "ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)\n\t"
#define D_ds_bperm(J) "ds_bpermute_b32 %0, %0, %0\n\ts_waitcnt lgkmcnt(0)\n\t"
#define D_readlane_v(J) "v_readlane_b32 %[s0], %0, 7\n\tv_add_u32 %0, %[s0], %0\n\t"
#define D_cmp_cnd(J) "v_cmp_lt_u32_e64 %[m0], %[a], %0\n\tv_cndmask_b32_e64 %0, %0, %[a], %[m0]\n\t"
#define D_s_add(J) "s_add_u32 %[s0], %[s0], %[k]\n\t"
#define D_s_ff1_lshl(J) "s_ff1_i32_b64 %[s0], %[m0]\n\ts_lshl_b64 %[m0], %[m0], %[s0]\n\t"
#define D_readlane_salu_v(J) "v_readlane_b32 %[s0], %0, 7\n\ts_add_u32 %[s0], %[s0], %[k]\n\tv_add_u32 %0, %[s0], %0\n\t"

#define T_V(NAME) KERNEL(NAME, REP64(M_##NAME))
#define T_L(NAME) KERNEL(NAME, REP64(M_##NAME) LDS_TAIL)
#define T_M4(NAME) KERNEL(NAME, REP64M4(M_##NAME))
#define T_D(NAME) KERNEL(dep_##NAME, REP64(D_##NAME))

T_V(fma) T_V(add) T_V(xor) T_V(lshl) T_V(mov) T_V(perm) T_V(movdpp_shr1) T_V(movdpp_bcast15) T_V(movdpp_bcast31)
T_V(movdpp_waveshr) T_V(ordpp_shr1) T_V(adddpp_shr1) T_V(cndmask_sgpr) T_V(cndmask_vcc) T_V(alignbyte) T_V(alignbit)
T_V(and_or) T_V(lshl_or) T_V(lshl_add) T_V(add3) T_V(bfi) T_V(mul24_sdwa) T_V(add_sdwa) T_V(mul24) T_V(mad24) T_V(mul_lo)
T_V(mul_hi) T_V(bfe) T_V(ffbl) T_V(ffbh) T_V(bfrev) T_V(bcnt) T_V(mbcnt) T_V(sad_u8) T_V(dot4) T_V(cmp_vcc) T_V(readlane)
T_V(readlane_sidx) T_V(readfirstlane) T_V(writelane) T_V(swap)
T_M4(cmp_sgpr)
T_V(and) T_V(or) T_V(sub) T_V(subrev) T_V(lshr) T_V(lshl_const) T_V(min) T_V(max) T_V(not) T_V(or3) T_V(xad) T_V(and_sgpr)
T_V(add_sgpr) T_V(add_lit) T_V(add_inline) T_V(add_e64) T_V(add_co) T_V(addc_co) T_V(cndmask_e64_vcc) T_V(pk_add_u16)
T_V(add_u16) T_V(and_sdwa) T_V(mov_sdwa) T_V(mul_f32)
KERNEL(cmp_cnd_vcc, REP32(M_cmp_cnd_vcc))
KERNEL(cmp_cnd_sgpr, REP32(M_cmp_cnd_sgpr))
T_L(ds_read_b64) T_L(ds_read_b128) T_L(ds_write_b64)
KERNEL(branch_not_taken, "s_cmp_eq_u32 %[k], %[k]\n\t" M_branch_not_taken(0) M_branch_not_taken(0) M_branch_not_taken(0) M_branch_not_taken(0) M_branch_not_taken(0) M_branch_not_taken(0) M_branch_not_taken(0) M_branch_not_taken(0) M_branch_not_taken(0) M_branch_not_taken(0) M_branch_not_taken(0) M_branch_not_taken(0) M_branch_not_taken(0) M_branch_not_taken(0) M_branch_not_taken(0) M_branch_not_taken(0) "Lend_%=:\n\t")
KERNEL(branch_taken, "s_cmp_eq_u32 %[k], %[k]\n\t" M_branch_taken(0) M_branch_taken(1) M_branch_taken(2) M_branch_taken(3) M_branch_taken(4) M_branch_taken(5) M_branch_taken(6) M_branch_taken(7) M_branch_taken(8) M_branch_taken(9) M_branch_taken(10) M_branch_taken(11) M_branch_taken(12) M_branch_taken(13) M_branch_taken(14) M_branch_taken(15))
KERNEL(global_load, REP64(M_global_load) "s_waitcnt vmcnt(0)\n\t")
T_L(ds_read_u8) T_L(ds_read_u16) T_L(ds_read_b32) T_L(ds_write_b8) T_L(ds_write_b16) T_L(ds_write_b32) T_L(ds_or_b32)
T_L(ds_add_u32) T_L(ds_bpermute) T_L(ds_permute) T_L(ds_swizzle)
T_V(s_add) T_V(s_mov) T_V(s_nop) T_V(s_cselect) T_V(s_mul) T_V(s_bcnt64) T_V(s_ff1_64)
T_M4(s_and64) T_M4(s_andn2_64) T_M4(s_lshl64) T_M4(s_bfm64) T_M4(s_bitset0)
KERNEL(mix_add_sadd, REP32(M_mix_add_sadd))
KERNEL(mix_perm_sand, REP32(M_mix_perm_sand))
KERNEL(mix_add_readlane, REP32(M_mix_add_readlane))
KERNEL(mix_add_dswrite, REP32(M_mix_add_dswrite) LDS_TAIL)
KERNEL(mix_add_snop, REP32(M_mix_add_snop))
KERNEL(mix3, REP32(M_mix3) LDS_TAIL)
T_D(add) T_D(perm) T_D(movdpp) T_D(ordpp) T_D(mul_lo) T_D(ds_read) T_D(ds_bperm) T_D(readlane_v) T_D(cmp_cnd) T_D(s_add)
T_D(s_ff1_lshl) T_D(readlane_salu_v)

typedef void (*kern_t)(Stamp*, unsigned*, int);
struct Test { const char* name; kern_t fn; int inst_per_iter; const char* note; };
#define E(NAME, NOTE) {#NAME, k_##NAME, 64, NOTE}

static Test tests[] = {
    E(fma, "v_fma_f32: the guide's 2-cycle yardstick"), E(add, "v_add_u32"), E(xor, "v_xor_b32"), E(lshl, "v_lshlrev_b32"),
    E(mov, "v_mov_b32"), E(perm, "v_perm_b32 (automaton composition)"),
    E(movdpp_shr1, "v_mov_b32_dpp row_shr:1"), E(movdpp_bcast15, "v_mov_b32_dpp row_bcast:15"),
    E(movdpp_bcast31, "v_mov_b32_dpp row_bcast:31"), E(movdpp_waveshr, "v_mov_b32_dpp wave_shr:1"),
    E(ordpp_shr1, "v_or_b32_dpp row_shr:1"), E(adddpp_shr1, "v_add_u32_dpp row_shr:1"),
    E(cndmask_sgpr, "v_cndmask_b32_e64 with an SGPR-pair selector (sel())"), E(cndmask_vcc, "v_cndmask_b32_e32 vcc"),
    E(alignbyte, "v_alignbyte_b32"), E(alignbit, "v_alignbit_b32"), E(and_or, "v_and_or_b32"), E(lshl_or, "v_lshl_or_b32"),
    E(lshl_add, "v_lshl_add_u32"), E(add3, "v_add3_u32"), E(bfi, "v_bfi_b32"),
    E(mul24_sdwa, "v_mul_u32_u24_sdwa src0_sel:WORD_1 (hash)"), E(mul24, "v_mul_u32_u24"), E(mad24, "v_mad_u32_u24"),
    E(mul_lo, "v_mul_lo_u32"), E(mul_hi, "v_mul_hi_u32"), E(bfe, "v_bfe_u32"), E(ffbl, "v_ffbl_b32"), E(ffbh, "v_ffbh_u32"),
    E(bfrev, "v_bfrev_b32"), E(bcnt, "v_bcnt_u32_b32"), E(mbcnt, "v_mbcnt_lo_u32_b32"), E(sad_u8, "v_sad_u8"),
    E(dot4, "v_dot4_u32_u8"), E(add_sdwa, "v_add_u32_sdwa src0_sel:BYTE_1"),
    E(and, "v_and_b32"), E(or, "v_or_b32"), E(sub, "v_sub_u32"), E(subrev, "v_subrev_u32"), E(lshr, "v_lshrrev_b32"),
    E(lshl_const, "v_lshlrev_b32 by an inline constant"), E(min, "v_min_u32"), E(max, "v_max_u32"), E(not, "v_not_b32"),
    E(or3, "v_or3_b32"), E(xad, "v_xad_u32"), E(and_sgpr, "v_and_b32 with an SGPR source"),
    E(add_sgpr, "v_add_u32 with an SGPR source"), E(add_lit, "v_add_u32 with a 32-bit literal"),
    E(add_inline, "v_add_u32 with an inline constant"), E(add_e64, "v_add_u32 in VOP3 encoding"),
    E(add_co, "v_add_co_u32 (writes vcc)"), E(addc_co, "v_addc_co_u32 (reads and writes vcc)"),
    E(cndmask_e64_vcc, "v_cndmask_b32_e64 with vcc named as the selector"), E(pk_add_u16, "v_pk_add_u16"),
    E(add_u16, "v_add_u16"), E(and_sdwa, "v_and_b32_sdwa"), E(mov_sdwa, "v_mov_b32_sdwa"), E(mul_f32, "v_mul_f32"),
    E(cmp_cnd_vcc, "32 x (v_cmp_lt_u32_e32 vcc ; v_cndmask_b32_e32 vcc): the compiler's select"),
    E(cmp_cnd_sgpr, "32 x (v_cmp_lt_u32_e64 sgpr ; v_cndmask_b32_e64 sgpr)"),
    E(ds_read_b64, "ds_read_b64"), E(ds_read_b128, "ds_read_b128"), E(ds_write_b64, "ds_write_b64"),
    E(global_load, "global_load_dword, 64 in flight, L2-resident 16 KiB"),
    {"branch_not_taken", k_branch_not_taken, 17, "s_cmp + 16 x s_cbranch_scc0 that falls through"},
    {"branch_taken", k_branch_taken, 17, "s_cmp + 16 x s_cbranch_scc1 taken (to the next instruction)"},
    E(cmp_sgpr, "v_cmp_lt_u32_e64 -> SGPR pair"), E(cmp_vcc, "v_cmp_lt_u32_e32 -> vcc"),
    E(readlane, "v_readlane_b32 constant lane"), E(readlane_sidx, "v_readlane_b32 SGPR lane"),
    E(readfirstlane, "v_readfirstlane_b32"), E(writelane, "v_writelane_b32"), E(swap, "v_swap_b32"),
    E(ds_read_u8, "ds_read_u8, lane-linear dwords"), E(ds_read_u16, "ds_read_u16"), E(ds_read_b32, "ds_read_b32"),
    E(ds_write_b8, "ds_write_b8"), E(ds_write_b16, "ds_write_b16"),
    E(ds_write_b32, "ds_write_b32"), E(ds_or_b32, "ds_or_b32 (no return)"), E(ds_add_u32, "ds_add_u32 (no return)"),
    E(ds_bpermute, "ds_bpermute_b32"), E(ds_permute, "ds_permute_b32"), E(ds_swizzle, "ds_swizzle_b32"),
    E(s_add, "s_add_u32"), E(s_mov, "s_mov_b32"), E(s_nop, "s_nop 0"), E(s_cselect, "s_cselect_b32"), E(s_mul, "s_mul_i32"),
    E(s_and64, "s_and_b64"), E(s_andn2_64, "s_andn2_b64"), E(s_lshl64, "s_lshl_b64"), E(s_bcnt64, "s_bcnt1_i32_b64"),
    E(s_ff1_64, "s_ff1_i32_b64"), E(s_bfm64, "s_bfm_b64"), E(s_bitset0, "s_bitset0_b64"),
    E(mix_add_sadd, "32 x (v_add_u32 ; s_add_u32): do the two pipes issue side by side"),
    E(mix_perm_sand, "32 x (v_perm_b32 ; s_and_b32)"), E(mix_add_readlane, "32 x (v_add_u32 ; v_readlane_b32)"),
    E(mix_add_dswrite, "32 x (v_add_u32 ; ds_write_b16)"), E(mix_add_snop, "32 x (v_add_u32 ; s_nop 0)"),
    {"mix3", k_mix3, 96, "32 x (v_add_u32 ; s_add_u32 ; ds_write_b16), 96 counted"},
    E(dep_add, "dependent v_add_u32 chain (latency)"), E(dep_perm, "dependent v_perm_b32 chain"),
    E(dep_movdpp, "dependent v_mov_b32_dpp chain"), E(dep_ordpp, "dependent v_or_b32_dpp chain"),
    E(dep_mul_lo, "dependent v_mul_lo_u32 chain"), E(dep_ds_read, "ds_read_b32 -> address -> ds_read_b32 (LDS round trip)"),
    E(dep_ds_bperm, "dependent ds_bpermute_b32"),
    {"dep_readlane_v", k_dep_readlane_v, 128, "v_readlane_b32 -> v_add_u32 using that SGPR -> readlane (pairs counted as 2)"},
    {"dep_cmp_cnd", k_dep_cmp_cnd, 128, "v_cmp -> SGPR -> v_cndmask chain (pairs counted as 2)"},
    E(dep_s_add, "dependent s_add_u32 chain"),
    {"dep_s_ff1_lshl", k_dep_s_ff1_lshl, 128, "s_ff1_i32_b64 -> s_lshl_b64 chain (pairs counted as 2)"},
    {"dep_readlane_salu_v", k_dep_readlane_salu_v, 192, "v_readlane -> s_add -> v_add chain (triples counted as 3): the walk's hop"},
};

int main(int argc, char** argv) {
    const char* only = argc > 1 ? argv[1] : nullptr;
    int iters = argc > 2 ? atoi(argv[2]) : 2000;
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    Stamp* d_st;
    unsigned* d_sink;
    const int maxgrid = cus * 32;
    CK(hipMalloc(&d_st, sizeof(Stamp) * maxgrid));
    CK(hipMalloc(&d_sink, 4096));
    CK(hipMemset(d_sink, 0, 4096));
    std::vector<Stamp> st(maxgrid);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // warm the clock: ~1.5 s of the yardstick
    for (int i = 0; i < 150; ++i) hipLaunchKernelGGL(k_fma, dim3(cus * 16), dim3(64), 8192, 0, d_st, d_sink, 20000);
    CK(hipDeviceSynchronize());
    // the cost of the stamps and of an empty loop
    const int Ws[] = {1, 2, 3, 4, 6, 8};
    for (const Test& t : tests) {
        if (only && strcmp(only, "all") && !strstr(t.name, only)) continue;
        for (int W : Ws) {
            const int wpc = 4 * W;
            size_t lds = (160 * 1024 / wpc) / 1280 * 1280;       // whole allocation units, so that exactly wpc fit
            if (lds > 64 * 1024) lds = 64 * 1024;
            if (lds < 2048) lds = 2048;
            if (W == 8) lds = 4096;
            const int grid = cus * wpc;
            CK(hipFuncSetAttribute((const void*)t.fn, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
            hipLaunchKernelGGL(t.fn, dim3(grid), dim3(64), lds, 0, d_st, d_sink, iters / 4);   // warm-up
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(t.fn, dim3(grid), dim3(64), lds, 0, d_st, d_sink, iters);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(st.data(), d_st, sizeof(Stamp) * grid, hipMemcpyDeviceToHost));
            // waves per SIMD as placed: HW_ID bits: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
            std::map<unsigned, int> per_simd;
            for (int i = 0; i < grid; ++i) per_simd[((st[i].xcc & 0xf) << 16) | (st[i].hwid & 0xff30)]++;
            std::vector<double> cyc, clk;
            for (int i = 0; i < grid; ++i) {
                const int w = per_simd[((st[i].xcc & 0xf) << 16) | (st[i].hwid & 0xff30)];
                const double dt = double(st[i].t1 - st[i].t0), dr = double(st[i].r1 - st[i].r0);
                cyc.push_back(dt / (double(iters) * t.inst_per_iter * w));
                if (dr > 0) clk.push_back(dt / dr * 100.0);
            }
            std::sort(cyc.begin(), cyc.end());
            std::sort(clk.begin(), clk.end());
            int wmin = 1 << 30, wmax = 0;
            for (auto& kv : per_simd) { wmin = std::min(wmin, kv.second); wmax = std::max(wmax, kv.second); }
            const double mhz = clk.empty() ? 0 : clk[clk.size() / 2];
            const double wall = ms * 1e-3 * mhz * 1e6 * (cus * 4.0) / (double(grid) * iters * t.inst_per_iter);
            printf("{\"test\": \"%s\", \"waves_per_simd\": %d, \"placed_min_max\": [%d, %d], \"simds_used\": %zu, "
                   "\"cyc_per_inst_simd\": %.3f, \"cyc_p10_p90\": [%.3f, %.3f], \"wall_cyc_per_inst\": %.3f, "
                   "\"clock_mhz\": %.0f, \"ms\": %.3f, \"what\": \"%s\"}\n",
                   t.name, W, wmin, wmax, per_simd.size(), cyc[cyc.size() / 2], cyc[cyc.size() / 10],
                   cyc[cyc.size() * 9 / 10], wall, mhz, ms, t.note);
            fflush(stdout);
        }
    }
    return 0;
}
