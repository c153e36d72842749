// hd_device.hpp -- wave64 primitives shared by the gfx950 kernels.
//
// Everything here assumes ONE wavefront of 64 lanes per block of work
// (north_star: "one wavefront per block").  No 32-wide idiom anywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/hipdeflate_params.h"

namespace hd {

// CRC-32 folding tables, built once on the host (hd_api.hip) and kept in HBM;
// 8.25 KiB, L1/L2 resident on every CU.
//   T[k][v] : slicing-by-4 tables of the reflected IEEE polynomial
//   B[k][v] : "append 1024 zero bytes" to the state byte v << 8k (B16: 16 zero bytes)
//   K[q]    : x^(128 q) mod P, q = 0..63, for the final per-lane alignment
//   SL[i]   : the static litlen code of RFC 1951 3.2.6, ready to OR into an LSB-first stream:
//             bits 0..15 the codeword (bit-reversed) and, for lengths, its extra bits behind it, bits 16..20
//             the bit count.  i < 256: literal i; i = 256 + (len - 3): match length len.  One load per
//             token in the level-1 emit pass replaces ~40 instructions of slot arithmetic and bit reversal.
//   P2[j]   : "append 2^j zero bytes" (j < 24), SM[s][m-1] : "append m segments of zero bytes" for the three
//             segment sizes the encoder uses (4080, 8160, 0xff00; m = 1..16), all in the byte-sliced form of B:
//             hd_segment.hpp folds the CRC-32 of a member from its segments' with one lookup per segment.
struct CrcTables {
	uint32_t T[4][256];
	uint32_t B[4][256];           // "append 1024 zero bytes" (a piece), byte-sliced like P2 / SM
	uint32_t B16[4][256];         // "append 16 zero bytes" (a lane's slot)
	uint32_t BL[4][256];          // "append 1008 zero bytes" (the CHAIN form of CrcLanes::fold)
	uint32_t T16[16][256];        // slicing-by-16: T16[j][b] = byte b followed by j zero bytes
	uint32_t K[64];
	uint32_t SL[512];
	uint32_t P2[24][4][256];
	uint32_t SM[3][16][4][256];
};

// ---- in-kernel clock (diagnostic build only: EXTRA=-DHD_CLOCK_STAMPS) -----------------------------------------
// MI355X_MICROARCH.md "DVFS give-back" item 6: the clock a kernel really runs at is delta s_memtime (shader cycles)
// over delta s_memrealtime (100 MHz), stamped once around the whole wave.  The sums go to a buffer of their own that
// nothing in the kernels reads (hipdeflate_test_clock() copies and clears it); in the product build ClockStamp is
// empty and no stamp executes.
#ifdef HD_CLOCK_STAMPS
__device__ unsigned long long g_clk[4][4];      // [kernel]{ sum of cycles, sum of 100 MHz ticks, waves, - }
__device__ unsigned long long g_clk_mark[16];   // [0,8) sums of cycles from a wave's start to mark(i), [8,16) waves that passed it (tools/clock_stamps_lat.py)
struct ClockStamp {
	unsigned long long t0, r0;
	int k;
	__device__ __forceinline__ void mark(int i) const
	{
		unsigned long long t;
		asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
		if (threadIdx.x == 0) {
			atomicAdd(&g_clk_mark[i], t - t0);
			atomicAdd(&g_clk_mark[8 + i], 1ull);
		}
	}
	__device__ __forceinline__ ClockStamp(int kernel) : k(kernel)
	{
		asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0), "=s"(t0)::"memory");
	}
	__device__ __forceinline__ ~ClockStamp()
	{
		unsigned long long t1, r1;
		asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
		if (threadIdx.x == 0) {
			atomicAdd(&g_clk[k][0], t1 - t0);
			atomicAdd(&g_clk[k][1], r1 - r0);
			atomicAdd(&g_clk[k][2], 1ull);
		}
	}
};
#else
struct ClockStamp {
	__device__ __forceinline__ ClockStamp(int) {}
	__device__ __forceinline__ void mark(int) const {}
};
#endif
enum { HD_CLK_STATIC = 0, HD_CLK_DYNAMIC = 1, HD_CLK_INFLATE = 2, HD_CLK_PARSE = 3 };

// BC: bound_ctrl -- "a lane without a source reads 0" said by the instruction instead of by an old value of 0.  The same values; with it
// the v_mov that seeds the old value is not emitted where no mask is set (two per wave_incl_scan).  Taken where it measured as a gain
// (the level-1 kernel: +0.3 %), left where it did not (k_inflate: -1 %, the emit kernels: nothing) -- tools/r05_ab_dev.sh
template <int CTRL, int ROW_MASK, int BANK_MASK, bool BC = false>
__device__ __forceinline__ uint32_t dpp0(uint32_t v)
{
	// lanes whose DPP source is masked off or out of the row read 0
	return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, BANK_MASK, BC);
}

// inclusive prefix sum over the 64 lanes: 7 DPP adds, no LDS
template <bool BC = false>
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x)
{
	uint32_t v = x;
	v += dpp0<0x111, 0xf, 0xf, BC>(x);   // row_shr:1
	v += dpp0<0x112, 0xf, 0xf, BC>(x);   // row_shr:2
	v += dpp0<0x113, 0xf, 0xf, BC>(x);   // row_shr:3
	v += dpp0<0x114, 0xf, 0xe, BC>(v);   // row_shr:4, banks 1-3
	v += dpp0<0x118, 0xf, 0xc, BC>(v);   // row_shr:8, banks 2-3
	v += dpp0<0x142, 0xa, 0xf, BC>(v);   // row_bcast:15 -> rows 1,3
	v += dpp0<0x143, 0xc, 0xf, BC>(v);   // row_bcast:31 -> rows 2,3
	return v;
}

// ---- greedy-parse automaton scan -------------------------------------------
// A lane's transition is f(r) = r - 1 for r = 1..7 and f(0) = a (a = len - 1 of
// the token that would start here, 0 for a literal): eight bytes
// {a,0,1,2 | 3,4,5,6}.  (g o f)(r) = g[f(r)] is a byte-table lookup, i.e.
// v_perm_b32 with g as the 8-byte table {S0 = g.hi, S1 = g.lo} and f's bytes as
// selectors (selector 0..3 -> S1 bytes, 4..7 -> S0 bytes).
struct Fn8 {
	uint32_t lo, hi;
};

// Lanes that a DPP stage leaves without a source must see the identity map
// {0,1,2,3 | 4,5,6,7}.  The shifted value is read with "0 where there is no
// source" and OR-ed with a per-lane constant that is the identity exactly on
// those lanes: mov_dpp + or fold into one v_or_b32_dpp, where seeding the
// destination with the identity would cost an extra v_mov per dword and stage.
struct Fn8Ident {
	uint32_t lo[6], hi[6];          // per scan stage
	__device__ __forceinline__ void init(uint32_t lane)
	{
		const uint32_t r = lane & 15;
		const bool none[6] = { r < 1, r < 2, r < 4, r < 8, !((lane >> 4) & 1), lane < 32 };
#pragma unroll
		for (int k = 0; k < 6; k++) {
			lo[k] = none[k] ? 0x03020100u : 0u;
			hi[k] = none[k] ? 0x07060504u : 0u;
		}
	}
};

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ Fn8 fn8_dpp(Fn8 v, uint32_t id_lo, uint32_t id_hi)
{
	Fn8 r;
	r.lo = dpp0<CTRL, ROW_MASK, 0xf>(v.lo) | id_lo;
	r.hi = dpp0<CTRL, ROW_MASK, 0xf>(v.hi) | id_hi;
	return r;
}

// apply `first`, then `then`
__device__ __forceinline__ Fn8 fn8_compose(Fn8 then, Fn8 first)
{
	Fn8 r;
	r.lo = __builtin_amdgcn_perm(then.hi, then.lo, first.lo);
	r.hi = __builtin_amdgcn_perm(then.hi, then.lo, first.hi);
	return r;
}

// inclusive scan: result at lane l = f_l o ... o f_0
__device__ __forceinline__ Fn8 fn8_scan(Fn8 w, const Fn8Ident &id)
{
	w = fn8_compose(w, fn8_dpp<0x111, 0xf>(w, id.lo[0], id.hi[0]));   // row_shr:1
	w = fn8_compose(w, fn8_dpp<0x112, 0xf>(w, id.lo[1], id.hi[1]));   // row_shr:2
	w = fn8_compose(w, fn8_dpp<0x114, 0xf>(w, id.lo[2], id.hi[2]));   // row_shr:4
	w = fn8_compose(w, fn8_dpp<0x118, 0xf>(w, id.lo[3], id.hi[3]));   // row_shr:8
	w = fn8_compose(w, fn8_dpp<0x142, 0xa>(w, id.lo[4], id.hi[4]));   // row_bcast:15 -> rows 1,3
	w = fn8_compose(w, fn8_dpp<0x143, 0xc>(w, id.lo[5], id.hi[5]));   // row_bcast:31 -> rows 2,3
	return w;
}

// the same scan with the identity seeded by v_mov instead of the 12 registers
// of Fn8Ident (the dynamic-level kernels sit at their VGPR budget)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ Fn8 fn8_dpp_seeded(Fn8 v)
{
	Fn8 r;
	r.lo = (uint32_t)__builtin_amdgcn_update_dpp((int)0x03020100, (int)v.lo, CTRL, ROW_MASK, 0xf, false);
	r.hi = (uint32_t)__builtin_amdgcn_update_dpp((int)0x07060504, (int)v.hi, CTRL, ROW_MASK, 0xf, false);
	return r;
}

__device__ __forceinline__ Fn8 fn8_scan(Fn8 w)
{
	w = fn8_compose(w, fn8_dpp_seeded<0x111, 0xf>(w));
	w = fn8_compose(w, fn8_dpp_seeded<0x112, 0xf>(w));
	w = fn8_compose(w, fn8_dpp_seeded<0x114, 0xf>(w));
	w = fn8_compose(w, fn8_dpp_seeded<0x118, 0xf>(w));
	w = fn8_compose(w, fn8_dpp_seeded<0x142, 0xa>(w));
	w = fn8_compose(w, fn8_dpp_seeded<0x143, 0xc>(w));
	return w;
}

__device__ __forceinline__ Fn8 fn8_make(bool live, uint32_t a)
{
	Fn8 f;
	f.lo = live ? (0x02010000u | a) : 0x03020100u;   // not live: identity
	f.hi = live ? 0x06050403u : 0x07060504u;
	return f;
}

__device__ __forceinline__ uint32_t wave_xor_reduce(uint32_t v)
{
	for (int o = 32; o > 0; o >>= 1)
		v ^= (uint32_t)__shfl_xor((int)v, o, 64);
	return v;
}

__device__ __forceinline__ uint32_t readlane(uint32_t v, uint32_t l)
{
	return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l);
}

__device__ __forceinline__ uint32_t uniform(uint32_t v)
{
	return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

// ---- lane masks ---------------------------------------------------------------
// A condition that is ONE compare reaches scalar code as a v_cmp straight into an SGPR pair (__ballot of the
// compare); conditions are then combined with s_and / s_or on the 64-bit masks and come back to the lanes
// through v_cndmask with the mask as its selector.  (A __ballot of an already combined bool costs a
// v_cndmask 0/1 + v_cmp_ne on top, and `(mask >> lane) & 1` three more VALU instructions.)
// d = lane's bit of `mask` ? a : b
__device__ __forceinline__ uint32_t sel(uint64_t mask, uint32_t a, uint32_t b)
{
	uint32_t d;
	asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(d) : "v"(b), "v"(a), "s"(mask));
	return d;
}

// ---- hash -> table slot ---------------------------------------------------------
// HD_HASH_SLOT (hipdeflate_params.h) in four full-rate instructions; returns the BYTE offset of the 16-bit
// entry, 2 * slot.  k2 / k1 / e2 / m are the constants in registers (SDWA takes no literals).
struct HashConsts {
	uint32_t k1, k2, e2, m;
	__device__ __forceinline__ void init(uint32_t entries)
	{
		k1 = HD_HASH_K1;
		k2 = HD_HASH_K2;
		e2 = 2 * entries;
		m = 0xfffeu;
	}
};
__device__ __forceinline__ uint32_t hash_slot_addr(uint32_t v, const HashConsts &k)
{
	uint32_t t1, t, x, a;
	asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD"
	    : "=v"(t1) : "v"(v), "v"(k.k2));
	asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(t) : "v"(v), "v"(k.k1), "v"(t1));
	// ((t >> 16) * 2 entries >> 16) & ~1 = 2 * (((t >> 16) * entries) >> 16)
	asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD"
	    : "=v"(x) : "v"(t), "v"(k.e2));
	asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD"
	    : "=v"(a) : "v"(x), "v"(k.m));
	return a;
}

// the lazy levels' key of six bytes (HD_HASH_SLOT6): vh = bytes [p+4, p+8), of which the low two count; returns the
// BYTE offset of the dword bucket, 4 * slot
struct HashConsts6 {
	uint32_t k1, k2, k3, e4, m;
	__device__ __forceinline__ void init(uint32_t buckets)
	{
		k1 = HD_HASH_K1;
		k2 = HD_HASH_K2;
		k3 = HD_HASH_K3;
		e4 = 4 * buckets;
		m = 0xfffcu;
	}
};
__device__ __forceinline__ uint32_t hash_slot_addr6(uint32_t v, uint32_t vh, const HashConsts6 &k)
{
	uint32_t t1, t2, t, x, a;
	asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD"
	    : "=v"(t1) : "v"(v), "v"(k.k2));
	asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD"
	    : "=v"(t2) : "v"(vh), "v"(k.k3));
	asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(t) : "v"(v), "v"(k.k1), "v"(t1));
	t += t2;
	// ((t >> 16) * 4 buckets >> 16) & ~3 = 4 * (((t >> 16) * buckets) >> 16)
	asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD"
	    : "=v"(x) : "v"(t), "v"(k.e4));
	asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD"
	    : "=v"(a) : "v"(x), "v"(k.m));
	return a;
}

// ---- CRC-32 ---------------------------------------------------------------
__device__ __forceinline__ uint32_t crc_step4(const CrcTables *ct, uint32_t s, uint32_t d)
{
	s ^= d;
	return ct->T[3][s & 0xff] ^ ct->T[2][(s >> 8) & 0xff] ^ ct->T[1][(s >> 16) & 0xff] ^ ct->T[0][s >> 24];
}

// one of the byte-sliced "append zero bytes" operators (P2[j], SM[s][m]) applied to a CRC state
__device__ __forceinline__ uint32_t crc_shift(const uint32_t (*Z)[256], uint32_t s)
{
	return Z[0][s & 0xff] ^ Z[1][(s >> 8) & 0xff] ^ Z[2][(s >> 16) & 0xff] ^ Z[3][s >> 24];
}

__device__ __forceinline__ uint32_t crc_byte(const CrcTables *ct, uint32_t s, uint32_t b)
{
	return ct->T[0][(s ^ b) & 0xff] ^ (s >> 8);
}

// a(x) * b(x) mod P, reflected representation (bit 31 = x^0)
__device__ __forceinline__ uint32_t gf_mul(uint32_t a, uint32_t b)
{
	uint32_t p = 0;
#pragma unroll 4
	for (int i = 0; i < 32; i++) {
		p ^= b & (0u - ((a >> (31 - i)) & 1u));
		b = (b >> 1) ^ (0xEDB88320u & (0u - (b & 1u)));
	}
	return p;
}

// the CRC state `s` followed by n more bytes that another CRC covers: s * x^(8 n), by the P2 operators
__device__ inline uint32_t crc_append_bytes(const CrcTables *ct, uint32_t s, uint64_t n)
{
	for (int j = 0; n && j < 24; j++, n >>= 1)
		if (n & 1)
			s = crc_shift(ct->P2[j], s);
	for (; n; n--)                                    // 16 MiB units beyond 2^24 bytes
		s = crc_shift(ct->P2[23], crc_shift(ct->P2[23], s));
	return s;
}

// Per-lane strided CRC accumulator.  The input is consumed in 1 KiB pieces;
// lane l owns the 16-byte slot j = 64 k + l of piece k.  fold() is called once
// per piece with that slot's bytes, finish() aligns and combines the 64 lanes.
struct CrcLanes {
	uint32_t s;
	// first: the CRC covers bytes [first, n) -- first a multiple of 16 below 1024, n - first >= 16 unless first == 0;
	// the caller folds the slots below `first` as not full
	__device__ __forceinline__ void init(uint32_t lane, uint32_t n, uint32_t first = 0)
	{
		s = (lane == first / 16 && n >= 16) ? 0xffffffffu : 0u;
	}
	// CHAIN: the same value with four loads in flight instead of twenty (the fused dynamic kernel sits at its
	// register budget): skip 1008 zero bytes, then four dependent slicing-by-4 steps
	template <bool CHAIN = false>
	__device__ __forceinline__ void fold(const CrcTables *ct, uint32_t piece, bool full, uint4 v)
	{
		if (CHAIN) {
			if (full) {
				uint32_t t = s;
				if (piece)
					t = ct->BL[0][s & 0xff] ^ ct->BL[1][(s >> 8) & 0xff] ^ ct->BL[2][(s >> 16) & 0xff] ^ ct->BL[3][s >> 24];
				t = crc_step4(ct, t, v.x);
				t = crc_step4(ct, t, v.y);
				t = crc_step4(ct, t, v.z);
				s = crc_step4(ct, t, v.w);
			}
			return;
		}
		if (full) {
			// CRC is linear: the state moved on by a piece (or, in the first piece, by the slot's 16 bytes) XOR the
			// slot's own 16 bytes by slicing-by-16 -- twenty table loads that do not wait for one another.  (The
			// chain it replaces, skip 1008 bytes then four slicing-by-4 steps, was five dependent rounds of loads
			// from global memory, ~3000 cycles per piece per wave.)
			const uint32_t (*Z)[256] = piece ? ct->B : ct->B16;
			uint32_t t = Z[0][s & 0xff] ^ Z[1][(s >> 8) & 0xff] ^ Z[2][(s >> 16) & 0xff] ^ Z[3][s >> 24];
			const uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
			for (int k = 0; k < 4; k++)
				t ^= ct->T16[15 - 4 * k][w[k] & 0xff] ^ ct->T16[14 - 4 * k][(w[k] >> 8) & 0xff] ^
				     ct->T16[13 - 4 * k][(w[k] >> 16) & 0xff] ^ ct->T16[12 - 4 * k][w[k] >> 24];
			s = t;
		}
	}
	// tail = the < 16 bytes after the last full slot
	__device__ __forceinline__ uint32_t finish(const CrcTables *ct, uint32_t lane, uint32_t n,
						    const uint8_t *tail)
	{
		uint32_t n16 = n >> 4;
		uint32_t acc = 0;
		if (lane < n16) {
			uint32_t jl = lane + 64u * ((n16 - 1 - lane) >> 6);
			acc = gf_mul(ct->K[n16 - 1 - jl], s);
		}
		acc = wave_xor_reduce(acc);
		if (n16 == 0)
			acc = 0xffffffffu;
		for (uint32_t i = n16 << 4; i < n; i++)
			acc = crc_byte(ct, acc, tail[i - (n16 << 4)]);
		return ~acc;
	}
};

// Candidates inside a step (HD_INTRA_DIST): the nearest of the DIST lanes before this one
// that holds the same four bytes `v`, as a distance (0 = none).  wave_shr:1 chains hand every lane the
// value of lane - d; lanes below d see filler and are masked.
template <int DIST>
__device__ __forceinline__ uint32_t intra_step_distance(uint32_t v, uint32_t lane)
{
	uint32_t sh = v, best = 0;
#pragma unroll
	for (int d = 1; d <= DIST; d++) {
		sh = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sh, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
		best = (best == 0 && sh == v && lane >= (uint32_t)d) ? (uint32_t)d : best;
	}
	return best;
}

// ---- Adler-32 (RFC 1950) ----------------------------------------------------
// a = 1 + sum d_i, b = n + sum (n - i) d_i (mod 65521): every lane keeps the plain and
// the index-weighted byte sum of its 16-byte slots (v_sad_u8 / v_dot4_u32_u8); bytes a
// ragged slot holds beyond n are zero and add nothing.
struct AdlerLanes {
	uint32_t s1, s2;
	__device__ __forceinline__ void init() { s1 = s2 = 0; }
	__device__ __forceinline__ void fold(uint32_t piece, uint32_t lane, uint4 v)
	{
		uint32_t sum = __builtin_amdgcn_sad_u8(v.x, 0u, 0u);
		sum = __builtin_amdgcn_sad_u8(v.y, 0u, sum);
		sum = __builtin_amdgcn_sad_u8(v.z, 0u, sum);
		sum = __builtin_amdgcn_sad_u8(v.w, 0u, sum);                     // <= 4080
		uint32_t ws = __builtin_amdgcn_udot4(v.x, 0x03020100u, 0u, false);
		ws = __builtin_amdgcn_udot4(v.y, 0x07060504u, ws, false);
		ws = __builtin_amdgcn_udot4(v.z, 0x0b0a0908u, ws, false);
		ws = __builtin_amdgcn_udot4(v.w, 0x0f0e0d0cu, ws, false);          // sum k * d_k, <= 30600
		const uint32_t g = (piece * HD_PIECE + lane * 16) % 65521u;      // index of the slot's first byte
		s1 += sum;                                                        // < 2^32 for n < 2^28
		s2 = (s2 + g * sum + ws) % 65521u;
	}
	__device__ __forceinline__ uint32_t finish(uint32_t n)
	{
		const uint32_t S1 = readlane(wave_incl_scan(s1 % 65521u), 63) % 65521u;
		const uint32_t S2 = readlane(wave_incl_scan(s2), 63) % 65521u;
		const uint32_t nm = n % 65521u;
		const uint32_t a = (1u + S1) % 65521u;
		const uint32_t b = (nm + (uint32_t)(((uint64_t)nm * S1) % 65521u) + 65521u - S2) % 65521u;
		return (b << 16) | a;
	}
};

// ---- RFC 1951 3.2.5 slot arithmetic ----------------------------------------
// length 3..258 -> symbol - 257, extra bit count, extra value
__device__ __forceinline__ void len_slot(uint32_t len, uint32_t &sym, uint32_t &eb, uint32_t &ev)
{
	uint32_t l = len - 3;
	uint32_t e = (l < 8) ? 0 : (29 - __clz(l));          // ilog2(l) - 2
	sym = (l < 8) ? l : (4 + 4 * e + ((l >> e) & 3));
	eb = e;
	ev = l & ((1u << e) - 1);
	if (l == 255) { sym = 28; eb = 0; ev = 0; }
}

// offset 1..32768 -> symbol, extra bit count, extra value
__device__ __forceinline__ void off_slot(uint32_t off, uint32_t &sym, uint32_t &eb, uint32_t &ev)
{
	uint32_t d = off - 1;
	uint32_t e = (d < 4) ? 0 : (30 - __clz(d));          // ilog2(d) - 1
	sym = (d < 4) ? d : (2 * e + 2 + ((d >> e) & 1));
	eb = e;
	ev = d & ((1u << e) - 1);
}

} // namespace hd
