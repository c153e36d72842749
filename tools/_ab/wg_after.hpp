// hd_deflate_wg.hpp -- levels >= HD_WG_LEVEL, throughput form: the WORKGROUP parse (BASELINE config 5, "level-6-like").
//
// Replaces, for levels 3..9 (HD_WG_LEVEL), the matchfinder and the parser of libdeflate's greedy and lazy levels -- hc_matchfinder
// (lib/libdeflate/hc_matchfinder.h:183-338: hash chains of depth 35 over a 32 KiB window) and deflate_compress_lazy_generic
// (lib/libdeflate/deflate_compress.c:2606-2809) -- with what ONE WORKGROUP holds in a CU's LDS: the whole DEFLATE window
// (a 64 KiB ring, distances up to 32768) and 64 KiB of table (HD_WG_BUCKETS(level) x HD_WG_WAYS(level) 16-bit entries), shared by HD_WG_WAVES wavefronts
// (include/hipdeflate_params.h "WORKGROUP LEVELS" states the algorithm; oracle/hd_deflate_twin.c deflate_wg() is its serial
// statement and must give the same bytes).  Rounds 1-3 gave every wavefront a private ring and table: 8 KiB + 2560 two-way
// buckets was what a share of LDS held, and "level 6" came out at libdeflate-1's ratio.
//
// A block is cut into PIECES of HD_WG_CUT = 1024 bytes (16 steps of 64 positions) and no match crosses a piece boundary, so
// the parse of a piece -- which candidate wins, the lazy rule, which lanes start a token -- depends on nothing but what the
// table held when the piece's positions looked their buckets up.  The wavefronts:
//   * wavefront 15, the FILLER: moves the block through the ring one piece at a time, at most WG_AHEAD pieces ahead of the
//     oldest piece still in work, and folds the block's CRC-32 from the pieces as they pass;
//   * wavefronts 0..14, the PARSERS: each takes the next piece from a counter.  For a piece it hashes its 16 steps (any time
//     after the bytes are in the ring), then waits for the piece's TURN -- the one thing that has an order: a step's lanes
//     must read their buckets as the steps before left them -- and with the turn reads and rewrites the 16 x 64 buckets
//     (one ds_read_b64, two ds_write_b32 per step at four ways; the old entries of every lane stay in registers), hands the
//     turn to the next piece, and then verifies, resolves and walks its piece alone, beside fourteen others doing the same.
// WAYS (1, 2, 4 positions per bucket = candidates verified per position) and LAZY are template parameters: the level ladder
// (include/hipdeflate_params.h HD_WG_WAYS / HD_WG_LAZY).
// The first version of this file (round 4, earlier) dealt single STEPS round robin and passed two turns per step (table,
// parse merge); 12 GB/s.  With the cut the parse needs no turn at all, the table turn is passed once per KiB (it costs 2 %,
// measured by leaving it out), and a wavefront's piece is 16 steps of straight-line work: 118 GB/s at four ways.
//
// Output: per piece its tokens (at most one per byte: piece k's start at token k * HD_WG_CUT of the block's record) and
// { tokens, literals, matches below 9 bytes, longer matches }; per block { status, CRC-32 }.  The emit-only kernel of
// hd_deflate_dynamic.hpp cuts the DEFLATE blocks (libdeflate's observation test over the pieces' counts, deflate_compress.c:
// 2141-2218), counts the symbols and writes the member.  A block of any length is one stream.
#pragma once
#include "hd_deflate_dynamic.hpp"

namespace hd {

#ifndef HD_WG_OR_OFFSETS
#define HD_WG_OR_OFFSETS 0                     /* (measured: no gain on 1 MiB members, -0.8 % on BGZF blocks -- tools/r05_ab_files.sh; the saturating adds stay) */
#endif
#ifndef HD_WG_UNIFORM_EDGES
#define HD_WG_UNIFORM_EDGES 1
#endif
#ifndef HD_WG_UNALIGNED_LDS
#define HD_WG_UNALIGNED_LDS 0                    /* own and candidate bytes by ONE ds_read_b128 at a byte address (gfx950 reads LDS unaligned) instead of five dwords + four v_alignbyte */
#endif
struct __attribute__((packed, aligned(1))) wg_u4u { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(1))) wg_u2u { uint32_t x, y; };
#ifndef HD_WG_LDS_AT_ZERO
#define HD_WG_LDS_AT_ZERO 1
#endif
#ifndef HD_WG_LDS_AT_ZERO_BASE
#define HD_WG_LDS_AT_ZERO_BASE 16                /* (not 0: address 0 cast to a pointer is the null pointer to the compiler, and what is reached through it is dead code) */
#endif
constexpr uint32_t WG_NW = HD_WG_WAVES;
constexpr uint32_t WG_NP = WG_NW - 1;            // parsers
constexpr uint32_t WG_STEPS = HD_WG_CUT / 64;
constexpr uint32_t WG_AHEAD = 24;                // pieces the ring is filled ahead of the oldest piece in work (or of the table
                                                 // turn, whichever is older): piece k takes the place of piece k - 64 and a
                                                 // parser reads up to 32 pieces behind its own, so anything below 32 is safe
#ifndef HD_WG_SPIN_LIMIT
#define HD_WG_SPIN_LIMIT (1u << 16)
#endif
constexpr uint32_t WG_SPIN_LIMIT = HD_WG_SPIN_LIMIT;   // a turn that does not come (~40 ms of polls where a wait is microseconds): the block is
                                                 // given up (stored, valid, NOT the twin's bytes) and counted in a.stalls -- never a
                                                 // hang, never silent: hipdeflate_stall_count() (bench.py, the GPU suite and the fuzz
                                                 // tools assert 0; tests/test_gpu_parity.py runs a build with a limit of 0 to see it move)
static_assert(HD_WG_CUT == HD_PIECE && WG_STEPS == 16, "a piece of the parse is a piece of the ring");
static_assert(HD_WG_RING == 65536 && HD_WG_WINDOW == 32768 && HD_WG_VCAP == 16, "the kernel is written for this geometry");
constexpr uint32_t WG_TABLE_BYTES = 65536;      // ways x buckets x 2 at every level

struct WgLds {
	__attribute__((aligned(16))) uint32_t ring32[HD_WG_RING / 4 + 8];      // + 32 bytes that mirror the start: unaligned reads never wrap
	__attribute__((aligned(16))) uint32_t table[WG_TABLE_BYTES / 4 + 2];   // p mod 2^16 x WAYS per bucket, newest first (+ a spare bucket)
	uint32_t turn;                         // the piece whose table accesses may run
	uint32_t filled;                       // pieces in the ring
	uint32_t fail;
	uint32_t next;                         // the next piece to hand out
	uint32_t cur[WG_NW];                   // the piece wavefront w is working on (0xffffffff: none)
};
#define WG_BARRIER() asm volatile("" ::: "memory")
// LDS words that other wavefronts write are read and written through address-space-3 pointers (ds_read / ds_write, never
// flat_*).  The LDS executes the instructions of all wavefronts of the workgroup one at a time and a wavefront's in order:
// who sees a word that a wavefront stored sees everything that wavefront stored before it.
#define WG_LDS __attribute__((address_space(3)))
typedef volatile WG_LDS uint32_t *wg_word_p;

// wait until *word >= want (the words only grow); false if the workgroup has failed or the word does not come
__device__ __forceinline__ bool wg_wait(wg_word_p word, uint32_t want, wg_word_p fail)
{
	for (uint32_t spins = 0;; spins++) {
		const uint32_t v = uniform(*word);
		WG_BARRIER();
		if (v >= want)
			return true;
		if (uniform(*fail) || spins > WG_SPIN_LIMIT) {
			*fail = 1;
			return false;
		}
		if (want - v >= 3)
			__builtin_amdgcn_s_sleep(24);              // the turn is pieces away
		else
			__builtin_amdgcn_s_sleep(2);
	}
}

// first BIT at which two 16-byte strings differ, given the XOR of their dwords, capped at `cap_bits` (<= 128): v_ffbl_b32 of
// an equal dword is 0xffffffff and stays there through the saturating add, so the minimum is the first differing dword's.
// Nine instructions, pinned (the three adds as ORs since round 5's end): the compiler's own form of "first set bit or the next dword's" is a compare and a select per dword.
__device__ __forceinline__ uint32_t wg_common_bits(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t cap_bits, uint32_t k96)
{
	uint32_t g0, g1, g2, g3, t;
	asm("v_ffbl_b32 %0, %1" : "=v"(g0) : "v"(x0));
	asm("v_ffbl_b32 %0, %1" : "=v"(g1) : "v"(x1));
	asm("v_ffbl_b32 %0, %1" : "=v"(g2) : "v"(x2));
	asm("v_ffbl_b32 %0, %1" : "=v"(g3) : "v"(x3));
#if HD_WG_OR_OFFSETS
	// (v_ffbl_b32 gives 0..31 or 0xffffffff: OR-ing the dword's bit offset in is the add, leaves 0xffffffff alone, and is a VOP2
	// instruction at the fast issue rate where the saturating add is VOP3)
	(void)k96;
	asm("v_or_b32_e32 %0, 32, %1" : "=v"(g1) : "v"(g1));
	asm("v_or_b32_e32 %0, 64, %1" : "=v"(g2) : "v"(g2));
	asm("v_or_b32_e32 %0, 0x60, %1" : "=v"(g3) : "v"(g3));
#else
	asm("v_add_u32_e64 %0, %1, 32 clamp" : "=v"(g1) : "v"(g1));
	asm("v_add_u32_e64 %0, %1, 64 clamp" : "=v"(g2) : "v"(g2));
	asm("v_add_u32_e64 %0, %1, %2 clamp" : "=v"(g3) : "v"(g3), "s"(k96));
#endif
	asm("v_min3_u32 %0, %1, %2, %3" : "=v"(t) : "v"(g0), "v"(g1), "v"(g2));
	asm("v_min3_u32 %0, %1, %2, %3" : "=v"(t) : "v"(t), "v"(g3), "v"(cap_bits));
	return t;
}

typedef uint32_t wg_u32x16 __attribute__((ext_vector_type(16)));

// the blocks of a latency launch, from wherever the caller has them (a latency context: pinned host memory) into device
// memory, once, for the workgroups that share their parse: one workgroup per block, four 16-byte loads per thread in flight
constexpr uint32_t WG_STAGE_STRIDE = 65536;
__global__ __launch_bounds__(1024) void k_stage_in(DeflateArgs a, uint8_t *stage)
{
	const uint32_t bi = blockIdx.x, b = a.first + bi;
	const uint8_t *src = a.in + a.in_off[b];
	const uint32_t n = a.in_len[b] < WG_STAGE_STRIDE ? a.in_len[b] : WG_STAGE_STRIDE;
	const bool aligned = (((uintptr_t)src) & 15) == 0;
	uint4 v[4];
#pragma unroll
	for (uint32_t i = 0; i < 4; i++)
		v[i] = load_slot(src, n, 16 * i + (threadIdx.x >> 6), threadIdx.x & 63, aligned);
#pragma unroll
	for (uint32_t i = 0; i < 4; i++)
		if ((16 * i + (threadIdx.x >> 6)) * HD_PIECE < n)
			*(uint4 *)(stage + (uint64_t)bi * WG_STAGE_STRIDE + (16 * i + (threadIdx.x >> 6)) * HD_PIECE + 16 * (threadIdx.x & 63)) = v[i];
}

__device__ __forceinline__ uint64_t wg_uniform64(uint64_t v)
{
	return ((uint64_t)uniform((uint32_t)(v >> 32)) << 32) | uniform((uint32_t)v);
}

// WAYS: positions per bucket = candidates verified per position (1, 2 or 4); LAZY: the lazy rule (else greedy)
// BESIDE: the instantiation that runs with the emit kernel resident beside it (launch_wg): its LDS is passed at launch, its records go
// through the L2, its blocks raise flags; BESIDE = 0 is the kernel of rounds 4-5 to the instruction
template <int WAYS, int LAZY, int BESIDE = 0>
__global__ __launch_bounds__(64 * HD_WG_WAVES) __attribute__((amdgpu_waves_per_eu(4, BESIDE ? 5 : 4))) void k_parse_wg(DeflateArgs a)
{
	static_assert(WAYS == 1 || WAYS == 2 || WAYS == 4, "a bucket is 2, 4 or 8 bytes");
	// BESIDE: DYNAMIC shared memory, on purpose: with the 131 KB declared statically the compiler knows that one workgroup fills the CU
	// and pads the kernel's register count from 95 to 97 (-> 104 allocated) "so that no fifth wavefront fits a SIMD" -- which also
	// keeps anything ELSE off the SIMD that needs more than 96 registers (round 5: the emit-only kernel beside the parse,
	// profiles/r05_wg_beside.txt).  Passed at launch (launch_wg), the size is not the compiler's business and 95 stays 96.
	WgLds *Lraw;
	if constexpr (BESIDE) {
		extern __shared__ __attribute__((aligned(16))) uint8_t wg_lds_raw[];
#if HD_WG_LDS_AT_ZERO
		// (the kernel has no static LDS, so what it is given at launch starts at offset 0 of the workgroup's allocation, and the launch asks for
		// HD_WG_LDS_AT_ZERO_BASE bytes more than the structure: it lies at that CONSTANT offset.  Through the symbol every address is
		// `v_add_u32 v, 0, v` first -- the 0 being the symbol's value, which arrives too late to be folded: five vector instructions per step)
		(void)wg_lds_raw;
		Lraw = (WgLds *)(WG_LDS WgLds *)(uintptr_t)HD_WG_LDS_AT_ZERO_BASE;
#else
		Lraw = (WgLds *)wg_lds_raw;
#endif
	} else {
		__shared__ WgLds Ls;
		Lraw = &Ls;
	}
	WgLds &L = *Lraw;
	const uint32_t lane = threadIdx.x & 63, w = uniform(threadIdx.x >> 6);      // (the compiler must know that w is one value per wavefront)
	// a.wg_split workgroups share a block's parse (latency launches): workgroup q of SP takes the pieces [pfirst, plast) and,
	// to have the table the pieces in front of them leave, REPLAYS those pieces' table turns first (hashes + bucket stores,
	// no verify, no tokens: an eighth of a piece's work)
	const uint32_t SP = a.wg_split > 1 ? a.wg_split : 1u;
	const uint32_t bi = blockIdx.x / SP, q = blockIdx.x % SP, b = a.first + bi;
	const uint8_t *src = a.stage_in ? a.stage_in + (uint64_t)bi * WG_STAGE_STRIDE : a.in + a.in_off[b];
	const uint32_t n = a.in_len[b];
	const bool aligned = (((uintptr_t)src) & 15) == 0;
	const CrcTables *ct = a.ct;
	const SplitLayout lay = wg_layout(a.split_max);
	uint8_t *const rec = a.scratch + (uint64_t)bi * lay.bytes;
	uint32_t *const tok = (uint32_t *)rec;
	uint4 *const rec_piece = (uint4 *)(rec + lay.off_ntok);
	WG_LDS WgLds *const Lp = (WG_LDS WgLds *)&L;
	const wg_word_p vturn = (wg_word_p)&Lp->turn, vfilled = (wg_word_p)&Lp->filled, vfail = (wg_word_p)&Lp->fail,
			vcur = (wg_word_p)Lp->cur;
	// The records are sized by a.split_max: the longest block of the launch where the host knows the lengths (the
	// host-pointer calls, pipes, latency contexts, the per-block codecs: round 5 -- a block LONGER THAN ITS ROOM is coded and
	// goes through whenever its stream fits, as libdeflate_deflate's does, lib/zlibutil.c:179-192; rounds 4 refused it), the
	// slot where only the device knows them (hipdeflate_batch_deflate_dev: there a block longer than its slot is refused)
	const bool refused = n > a.split_max;
	const bool beside = BESIDE != 0;                 // the emit kernel runs beside this one and reads the records as the flags go up (launch_wg)
#if HD_BESIDE_PARSE_PRIO
	if constexpr (BESIDE != 0)
		__builtin_amdgcn_s_setprio(HD_BESIDE_PARSE_PRIO);
#endif
	const uint32_t npieces = refused ? 0u : (n + HD_WG_CUT - 1) / HD_WG_CUT;
	const uint32_t pfirst = npieces * q / SP, plast = npieces * (q + 1) / SP;    // (SP == 1: all of them)

	// ---- LDS: the table zero, the words ------------------------------------------------------------------------------
	for (uint32_t i = threadIdx.x; i < WG_TABLE_BYTES / 16; i += 64 * WG_NW)
		((uint4 *)L.table)[i] = make_uint4(0, 0, 0, 0);
	if (threadIdx.x < WG_NW)
		L.cur[threadIdx.x] = 0xffffffffu;
	// A block that fits the ring whole (every BGZF block; everything behind the per-block boundary) is brought in by ALL
	// sixteen wavefronts at once, four 16-byte loads per lane in flight, before anybody parses: the filler's one piece
	// ahead is one memory latency per KiB -- microseconds each when the block lies in the caller's pinned memory (the
	// hook, hip_deflate: 64 round trips over PCIe were a quarter of a lone block's parse) -- and nothing is ever replaced
	// in the ring, so there is no order to keep.  The filler then only folds the CRC-32 (from the ring) and parses too.
	const bool bulk = npieces <= HD_WG_RING / HD_PIECE;
	if (bulk) {
		uint4 v[HD_WG_RING / HD_PIECE / WG_NW];
#pragma unroll
		for (uint32_t i = 0; i < HD_WG_RING / HD_PIECE / WG_NW; i++)
			if (w + WG_NW * i < npieces)
				v[i] = load_slot(src, n, w + WG_NW * i, lane, aligned);
#pragma unroll
		for (uint32_t i = 0; i < HD_WG_RING / HD_PIECE / WG_NW; i++)
			if (w + WG_NW * i < npieces) {
				const uint32_t o = (w + WG_NW * i) * HD_PIECE + 16 * lane;
				*(uint4 *)((uint8_t *)L.ring32 + o) = v[i];
				if (o < 32)                        // the mirror behind the ring's end
					*(uint4 *)((uint8_t *)L.ring32 + HD_WG_RING + o) = v[i];
			}
	}
	if (threadIdx.x == 0) {
		L.turn = L.next = 0;
		// (beside, SPAN: a gate in front of this parse gave up -- the records this block would overwrite may still be wanted: it is given
		// up like a block whose turn did not come, before a token is written)
		L.fail = (BESIDE != 0 && a.poison && __hip_atomic_load(a.poison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) ? 1u : 0u;
		L.filled = bulk ? npieces : 0u;
		L.table[WG_TABLE_BYTES / 4] = L.table[WG_TABLE_BYTES / 4 + 1] = 0;
	}
	__syncthreads();
	const bool poisoned = BESIDE != 0 && uniform(L.fail) != 0;      // (nothing is read, nothing written: the emit kernel stores the block)

	uint32_t crcv = 0;
	if (w == WG_NP && !poisoned) {
		// ================= the filler =====================================================================================
		CrcLanes crc;
		crc.init(lane, n);
		uint4 pend = make_uint4(0, 0, 0, 0);
		if (bulk && q) {
			// (the block's CRC-32 is workgroup 0's)
		} else if (bulk) {
			for (uint32_t k = 0; k < npieces; k++)
				crc.fold(ct, k, k * HD_PIECE + 16 * lane + 16 <= n, *(const uint4 *)((const uint8_t *)L.ring32 + k * HD_PIECE + 16 * lane));
		} else if (npieces) {
			pend = load_slot(src, n, 0, lane, aligned);
		}
		for (uint32_t k = 0; !bulk && k < npieces; k++) {
			// piece k takes the place of piece k - 64, and a parser reads up to 32 pieces behind its own: k stays within
			// WG_AHEAD of the oldest piece still in work (the table turn counts as one: its holder is about to start)
			if (k > WG_AHEAD) {
				bool ok = true;
				for (uint32_t spins = 0;; spins++) {
					uint32_t v = lane < WG_NW ? vcur[lane] : 0xffffffffu;
					const uint32_t tn = uniform(*vturn);
					WG_BARRIER();
					v = min(v, (uint32_t)__shfl_xor((int)v, 8, 64));
					v = min(v, (uint32_t)__shfl_xor((int)v, 4, 64));
					v = min(v, (uint32_t)__shfl_xor((int)v, 2, 64));
					v = min(v, (uint32_t)__shfl_xor((int)v, 1, 64));
					const uint32_t oldest = min(uniform(v), tn);
					if (oldest + WG_AHEAD >= k)
						break;
					if (uniform(*vfail) || spins > WG_SPIN_LIMIT) {
						*vfail = 1;
						ok = false;
						break;
					}
					__builtin_amdgcn_s_sleep(8);
				}
				if (!ok)
					break;
			}
			const uint4 v = pend;
			if (k + 1 < npieces)
				pend = load_slot(src, n, k + 1, lane, aligned);
			const uint32_t o = (k * HD_PIECE + 16 * lane) & (HD_WG_RING - 1);
			*(uint4 *)((uint8_t *)L.ring32 + o) = v;
			if (o < 32)                        // the mirror behind the ring's end
				*(uint4 *)((uint8_t *)L.ring32 + HD_WG_RING + o) = v;
			WG_BARRIER();
			if (lane == 0)
				*vfilled = k + 1;
			WG_BARRIER();
			crc.fold(ct, k, k * HD_PIECE + 16 * lane + 16 <= n, v);
		}
		crcv = crc.finish(ct, lane, n, src + (n & ~15u));
	}
	if ((w != WG_NP || bulk) && !poisoned) {
		// ================= a parser (bulk: the filler too, once its CRC is folded) =====================================
		HashConsts6 hk;
		hk.init(WG_TABLE_BYTES / 4);               // byte offset of a bucket of 2 WAYS bytes: (2 WAYS) * slot, below 64 KiB
		hk.m = 0x10000u - 2 * WAYS;
		WG_LDS uint8_t *const tab8 = (WG_LDS uint8_t *)Lp->table;
		uint32_t k96, kfffc;
		asm volatile("s_movk_i32 %0, 96" : "=s"(k96));            // (an SGPR on purpose: VOP3 takes no literal)
		asm volatile("v_mov_b32 %0, 0xfffc" : "=v"(kfffc));        // (a VGPR on purpose: SDWA takes no literal)
		const uint8_t *const ring8 = (const uint8_t *)L.ring32;
		// Pieces are handed out by a counter, not dealt round robin: three SIMDs carry four parsers and one carries three and
		// the filler, and a parser that gets a larger share of its SIMD takes more pieces.  Whoever holds the lowest piece in
		// work never waits for a higher one, so the hand-out cannot lock up.
		uint32_t deal_round = 0;
		(void)deal_round;
		for (;;) {
			uint32_t j = 0;
#ifdef HD_WG_EXP_STATIC_DEAL                        /* experiment (tools/exp_wg_variants.sh): pieces dealt round robin */
			j = w + WG_NP * deal_round++;
#else
			if (lane == 0)
				j = __hip_atomic_fetch_add((WG_LDS uint32_t *)&Lp->next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			j = uniform(j);
#endif
			if (lane == 0)
				vcur[w] = j < plast ? j : 0xffffffffu;
			WG_BARRIER();
			if (j >= plast)
				break;
			const uint32_t P0 = j * HD_WG_CUT;
			const uint32_t pend = n - P0 < HD_WG_CUT ? n : P0 + HD_WG_CUT;         // the piece's end
			const uint32_t nst = (pend - P0 + 63) >> 6;
			// the piece's bytes and the five behind it (the key of its last positions)
			if (!wg_wait(vfilled, j + 2 < npieces ? j + 2 : npieces, vfail))
				break;
			// ---- hashes of the 16 steps (no turn needed) ------------------------------------------------------------
			const bool all_keyed = pend + HD_LAZY_KEY_BYTES - 1 <= n && nst == WG_STEPS;
			wg_u32x16 ha;
#pragma unroll
			for (int t = 0; t < (int)WG_STEPS; t++) {
				const uint32_t p = P0 + 64 * t + lane;
#if HD_WG_UNALIGNED_LDS
				const wg_u2u kv = *(const wg_u2u *)((const uint8_t *)L.ring32 + (p & (HD_WG_RING - 1)));
				const uint32_t v = kv.x, vh = kv.y;
#else
				const uint32_t *q = L.ring32 + ((p & (HD_WG_RING - 1)) >> 2);
				const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
				const uint32_t v = __builtin_amdgcn_alignbyte(d1, d0, p), vh = __builtin_amdgcn_alignbyte(d2, d1, p);   // (v_alignbyte_b32 reads bits [1:0] of its shift: tools/isa_probe.hip)
#endif
				ha[t] = hash_slot_addr6(v, vh, hk);
			}
			// ---- the turn: 16 steps of buckets, in order --------------------------------------------------------------
#ifndef HD_WG_EXP_NO_TURN                           /* experiment (timing only: the bytes then depend on the race): no table turn */
			if (!wg_wait(vturn, j, vfail))
				break;
#endif
			__builtin_amdgcn_s_setprio(3);
			wg_u32x16 cx, cy;
			// one step's buckets: read, { itself, the WAYS - 1 newest before } written back; of the lanes of a step that share
			// a bucket the HIGHEST keeps each store (the LDS-order probe of ctx_init checks ds_write_b16 / b32), so a bucket of
			// eight bytes is two dword stores, both that lane's
			auto bucket = [&](uint32_t h, uint32_t p, uint32_t &ox, uint32_t &oy) {
				if (WAYS == 4) {
					const unsigned long long o64 = *(WG_LDS const volatile unsigned long long *)(tab8 + h);   // ds_read_b64
					ox = (uint32_t)o64;
					oy = (uint32_t)(o64 >> 32);
					*(wg_word_p)(tab8 + h) = (ox << 16) | (p & 0xffffu);
					*(wg_word_p)(tab8 + h + 4) = __builtin_amdgcn_alignbit(oy, ox, 16);
				} else if (WAYS == 2) {
					ox = *(wg_word_p)(tab8 + h);
					*(wg_word_p)(tab8 + h) = (ox << 16) | (p & 0xffffu);
				} else {
					ox = *(volatile WG_LDS uint16_t *)(tab8 + h);
					*(volatile WG_LDS uint16_t *)(tab8 + h) = (uint16_t)p;
				}
			};
			if (all_keyed) {
#pragma unroll
				for (int t = 0; t < (int)WG_STEPS; t++) {
					uint32_t ox = 0, oy = 0;
					bucket(ha[t], P0 + 64 * t + lane, ox, oy);
					cx[t] = ox;
					cy[t] = oy;
				}
			} else {
#pragma unroll
				for (int t = 0; t < (int)WG_STEPS; t++) {
					const uint32_t p = P0 + 64 * t + lane;
					uint32_t ox = 0, oy = 0;
					if (p + HD_LAZY_KEY_BYTES <= n)
						bucket(ha[t], p, ox, oy);
					cx[t] = ox;
					cy[t] = oy;
				}
			}
			WG_BARRIER();
			if (lane == 0)
				*vturn = j + 1;
			WG_BARRIER();
			__builtin_amdgcn_s_setprio(0);
			if (j < pfirst)
				continue;                              // a replayed piece: its stores are in the table, another workgroup parses it

			// ---- the piece: verify, lazy rule, walk, tokens -- nothing here waits for another wavefront ----------------
			uint32_t E = P0;                           // first position no token covers yet
			uint32_t cnt = 0, c_lit = 0, c_long = 0;
			uint32_t *const ptok = tok + P0;           // the piece's tokens
			for (uint32_t t = 0; t < nst; t++) {
				const uint32_t S = P0 + 64 * t, p = S + lane;
				const uint32_t lanes = pend - S < 64 ? pend - S : 64;
				const uint64_t lanem = lanes == 64 ? ~0ull : (1ull << lanes) - 1;
				// own 16 bytes
#if HD_WG_UNALIGNED_LDS
				const wg_u4u ov = *(const wg_u4u *)((const uint8_t *)L.ring32 + (p & (HD_WG_RING - 1)));
				const uint32_t o0 = ov.x, o1 = ov.y, o2 = ov.z, o3 = ov.w;
#else
				const uint32_t *q = L.ring32 + ((p & (HD_WG_RING - 1)) >> 2);
				const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4];
				const uint32_t sh = p;                           // (v_alignbyte_b32 reads bits [1:0] only)
				const uint32_t o0 = __builtin_amdgcn_alignbyte(d1, d0, sh), o1 = __builtin_amdgcn_alignbyte(d2, d1, sh),
					       o2 = __builtin_amdgcn_alignbyte(d3, d2, sh), o3 = __builtin_amdgcn_alignbyte(d4, d3, sh);
#endif
#if HD_WG_UNIFORM_EDGES
				// (two per-lane values that are the same in every lane but at a block's edges -- its last positions, a piece's last
				// step: decided by the scalar unit, the vector instructions only where they differ)
				// (the empty asm statements: left to itself the compiler computes both sides and selects)
				uint64_t keyed = ~0ull;
				if (S + 63 + HD_LAZY_KEY_BYTES > n) {
					keyed = __ballot(p + HD_LAZY_KEY_BYTES <= n);
					asm volatile("" : "+s"(keyed));
				}
				uint32_t room8 = (uint32_t)HD_WG_VCAP << 3;                      // in bits (keyed lanes: p < pend)
				if (S + 63 + HD_WG_VCAP > pend) {
					room8 = min(pend - p, (uint32_t)HD_WG_VCAP) << 3;
					asm volatile("" : "+v"(room8));
				}
				const uint32_t lim = min(p, (uint32_t)HD_WG_WINDOW);              // (one instruction either way: no branch for it)
#else
				const uint64_t keyed = __ballot(p + HD_LAZY_KEY_BYTES <= n);
				const uint32_t room8 = min(pend - p, (uint32_t)HD_WG_VCAP) << 3;  // in bits (keyed lanes: p < pend)
				const uint32_t lim = min(p, (uint32_t)HD_WG_WINDOW);
#endif
				// the byte before (runs; only inside the step): as long as the own bytes repeat it
				uint32_t best, dm1 = 0;                         // the best candidate's length, its distance - 1
				{
					// (bound_ctrl: the lane without a neighbour reads 0 -- said with an old value of 0 instead, every step pays a v_mov for it)
					const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)o0, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
					const uint32_t sp = __builtin_amdgcn_perm(prev, prev, 0u);      // its first byte, four times
					const uint32_t m = wg_common_bits(o0 ^ sp, o1 ^ sp, o2 ^ sp, o3 ^ sp, room8, k96) >> 3;
					best = sel(keyed & ~1ull, m, 0u);
				}
				// the bucket, newest first: the longest wins, the nearer on a tie.  An entry e is a ring offset; its distance
				// - 1 = (p - 1 - e) mod 2^16 in ONE 16-bit subtract with the entry's half picked by SDWA, and the entry is in
				// range when that is below min(p, 32768) (distance 0 wraps to 65535)
				const uint32_t cxt = cx[t], cyt = WAYS == 4 ? cy[t] : 0u;
				const uint32_t pm1 = p - 1;
#pragma unroll
				for (int k = 0; k < WAYS; k++) {
					const uint32_t cw = k < 2 ? cxt : cyt;
					uint32_t bm1, ea;
					if (k & 1) {
						asm("v_sub_u16_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "=v"(bm1) : "v"(pm1), "v"(cw));
						asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "=v"(ea) : "v"(cw), "v"(kfffc));
					} else {
						asm("v_sub_u16_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_0" : "=v"(bm1) : "v"(pm1), "v"(cw));
						asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "=v"(ea) : "v"(cw), "v"(kfffc));
					}
					const uint64_t ok = __ballot(bm1 < lim) & keyed;
#if HD_WG_UNALIGNED_LDS
					uint32_t eb;                                          // the entry as it is: a byte offset in the ring
					if (k & 1)
						asm("v_mov_b32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(eb) : "v"(cw));
					else
						asm("v_mov_b32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(eb) : "v"(cw));
					(void)ea;
					const wg_u4u cv = *(const wg_u4u *)((const uint8_t *)L.ring32 + eb);
					const uint32_t x0 = cv.x ^ o0, x1 = cv.y ^ o1, x2 = cv.z ^ o2, x3 = cv.w ^ o3;
#else
					const uint32_t *c = (const uint32_t *)((const uint8_t *)L.ring32 + ea);
					const uint32_t c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3], c4 = c[4];
					const uint32_t cs = (k & 1) ? cw >> 16 : cw;          // (v_alignbyte_b32 reads bits [1:0] of its shift)
					const uint32_t x0 = __builtin_amdgcn_alignbyte(c1, c0, cs) ^ o0, x1 = __builtin_amdgcn_alignbyte(c2, c1, cs) ^ o1,
						       x2 = __builtin_amdgcn_alignbyte(c3, c2, cs) ^ o2, x3 = __builtin_amdgcn_alignbyte(c4, c3, cs) ^ o3;
#endif
					const uint32_t m = wg_common_bits(x0, x1, x2, x3, room8, k96) >> 3;
					const uint64_t better = __ballot(m > best) & ok;
					best = sel(better, m, best);
					dm1 = sel(better, bm1, dm1);
				}
				const uint32_t dist = dm1 + 1;
				const uint64_t candm = __ballot(best >= HD_WG_MIN_LEN) & lanem;
				// (the lazy rule and the jump read `best` itself: a candidate's length is its best, and where the rule looks at a lane that
				// is none -- the neighbour -- a best below five loses every comparison a zero would: one select less)
				const uint32_t clen = best;
				// ---- the lazy rule on the lane to the right (deflate_compress.c:2723-2726, lengths capped at 16) ------
				uint64_t defer = 0;
				if (LAZY) {
					const uint32_t clen_r = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)clen, 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
					const uint32_t dist_r = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)dist, 0x130, 0xf, 0xf, true);   // (lane 63: 0, and `| 1` below)
					// (a distance is >= 1, so its leading zeros are those of `dist | 1` -- the twin's form -- without the OR; lane 63's
					// neighbour is nobody and its verdict is masked out below)
					// (v_ffbh_u32 as it is: __clz adds a fix-up for 0, which no distance is)
					uint32_t lz_r, lz;
					asm("v_ffbh_u32 %0, %1" : "=v"(lz_r) : "v"(dist_r));
					asm("v_ffbh_u32 %0, %1" : "=v"(lz) : "v"(dist));
					const int gain = (int)((((uint32_t)clen_r - (uint32_t)clen) << 2) + lz_r) - (int)lz;
					defer = __ballot(clen_r >= clen) & __ballot(gain > 2) & candm & (lanem >> 1);
				}
				const uint64_t take = candm & ~defer;
				// matches of the whole verified span are extended when the walk takes them
				uint64_t capt = __ballot(best == HD_WG_VCAP) & take;
				uint32_t jmp = sel(take, clen, 1u);             // where the parse goes from a lane that starts a token
				// ---- the walk from E: a hop per match (a bit set, a lane read, an add) and one per run of literals (a mask), in
				// ISA -- the compiler's form of this loop was ~27 scalar instructions per hop, and the scalar pipe is as full as
				// the vector one here -----
				uint64_t starts = 0;
				uint32_t bb = E > S ? E - S : 0u;
				if (take == 0 && bb < lanes) {                   // (nothing but literals: random data)
					starts = lanem & ~((1ull << bb) - 1);
					bb = lanes;
				}
				while (bb < lanes) {
					uint32_t hop;
					capt = wg_uniform64(capt);
					const uint64_t take_u = wg_uniform64(take);
					starts = wg_uniform64(starts);
					bb = uniform(bb);
					uint64_t tk;
					uint32_t z;
					asm volatile("Lhd_wg_walk_%=:\n\t"
						     "s_bitcmp1_b64 %[take], %[b]\n\t"
						     "s_cbranch_scc0 Lhd_wg_lit_%=\n"
						     "Lhd_wg_match_%=:\n\t"
						     "s_bitcmp1_b64 %[capt], %[b]\n\t"
						     "s_cbranch_scc1 Lhd_wg_walk_out_%=\n\t"
						     "s_bitset1_b64 %[st], %[b]\n\t"
						     "v_readlane_b32 %[hop], %[jmp], %[b]\n\t"
						     "s_add_u32 %[b], %[b], %[hop]\n\t"
						     "s_cmp_lt_u32 %[b], %[lanes]\n\t"
						     "s_cbranch_scc1 Lhd_wg_walk_%=\n\t"
						     "s_branch Lhd_wg_walk_out_%=\n"
						     "Lhd_wg_lit_%=:\n\t"                      // a run of literals: up to the next match or the step's end
						     "s_lshr_b64 %[tk], %[take], %[b]\n\t"
						     "s_ff1_i32_b64 %[z], %[tk]\n\t"           // (-1: no match behind)
						     "s_sub_u32 %[hop], %[lanes], %[b]\n\t"
						     "s_min_u32 %[z], %[z], %[hop]\n\t"        // (< 64: a whole step of literals never comes here)
						     "s_bfm_b64 %[tk], %[z], %[b]\n\t"
						     "s_or_b64 %[st], %[st], %[tk]\n\t"
						     "s_add_u32 %[b], %[b], %[z]\n\t"
						     "s_cmp_lt_u32 %[b], %[lanes]\n\t"
						     "s_cbranch_scc1 Lhd_wg_match_%=\n"
						     "Lhd_wg_walk_out_%=:"
						     : [b] "+s"(bb), [st] "+s"(starts), [hop] "=&s"(hop), [tk] "=&s"(tk), [z] "=&s"(z)
						     : [capt] "s"(capt), [jmp] "v"(jmp), [lanes] "s"(lanes), [take] "s"(take_u)
						     : "scc");
					// (what an asm statement returns counts as divergent for the compiler: pinned)
					bb = uniform(bb);
					starts = wg_uniform64(starts);
					if (bb >= lanes)
						break;
					// lane bb: a match of the whole verified span, taken: to its full length, 64 bytes per pass by all lanes
					const uint32_t k = bb;
					uint32_t len = HD_WG_VCAP;
					const uint32_t D = readlane(dist, k), at = S + k;
					const uint32_t maxlen = pend - at < HD_MAX_MATCH ? pend - at : HD_MAX_MATCH;
					while (len < maxlen) {
						const uint32_t x = at + len + lane;
						const uint64_t ne = __ballot(ring8[x & (HD_WG_RING - 1)] != ring8[(x - D) & (HD_WG_RING - 1)]);
						const uint32_t adv = ne ? (uint32_t)__builtin_ctzll(ne) : 64u;
						len = len + adv < maxlen ? len + adv : maxlen;
						if (ne)
							break;
					}
					// (v_writelane_b32 with the lane select in M0: the one form that may name two scalar operands)
					asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(jmp) : "s"(len), "s"(k));
					capt &= ~(1ull << k);
				}
				E = S + bb;
				const uint32_t flen = jmp;                       // (a token's length on the lanes of `take`)
				// ---- the step's tokens ------------------------------------------------------------------------------
				const uint64_t long9 = __ballot(flen >= 9);
				// token words: bits 16..24 are the index into the emit kernel's litlen table -- a literal's byte, or
				// 256 + (length - 3) (HD_TOKEN_MATCH_TAG) -- and a match carries its distance - 1 below
				uint32_t litw;
				asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(litw) : "v"(16u), "v"(o0));
				const uint32_t tw = sel(take, (flen << 16) + dm1 + (HD_TOKEN_MATCH_TAG - (3u << 16)), litw);
				{
					const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(starts >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)starts, 0));
					// (the address as a scalar base -- the piece's tokens so far -- and a 32-bit lane offset: no 64-bit vector add)
					const uint32_t *const at_base = (const uint32_t *)wg_uniform64((uint64_t)(ptok + cnt));
					const uint32_t at_off = rank << 2;
					// (the store under exec = starts: the mask goes to exec as it is, not through a compare per lane)
					uint64_t saved;
					// (beside: the reader sits behind another XCD's L2 -- the store goes through this one's, sc1)
					if (beside)
						asm volatile("s_and_saveexec_b64 %0, %1\n\tglobal_store_dword %2, %3, %4 sc1\n\ts_mov_b64 exec, %0"
							     : "=&s"(saved) : "s"(starts), "v"(at_off), "v"(tw), "s"(at_base) : "memory");
					else
						asm volatile("s_and_saveexec_b64 %0, %1\n\tglobal_store_dword %2, %3, %4\n\ts_mov_b64 exec, %0"
							     : "=&s"(saved) : "s"(starts), "v"(at_off), "v"(tw), "s"(at_base) : "memory");
				}
				cnt += (uint32_t)__popcll(starts);
				c_lit += (uint32_t)__popcll(starts & ~take);
				c_long += (uint32_t)__popcll(starts & take & long9);
			}
			if (lane == 0) {
				if (beside) {
					uint32_t *rp = (uint32_t *)&rec_piece[j];
					__hip_atomic_store(rp + 0, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					__hip_atomic_store(rp + 1, c_lit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					__hip_atomic_store(rp + 2, cnt - c_lit - c_long, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					__hip_atomic_store(rp + 3, c_long, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				} else {
					rec_piece[j] = make_uint4(cnt, c_lit, cnt - c_lit - c_long, c_long);
				}
			}
		}
	}
	// (beside: every wavefront's stores -- written through this XCD's L2, sc1 -- have been acknowledged before the barrier)
	if (beside)
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	if (w == WG_NP && lane == 0) {
		uint32_t *m = (uint32_t *)(rec + lay.off_rec);
		const bool stalled = uniform(*vfail) != 0;
		if (q == 0) {
			if (beside) {
				__hip_atomic_store(&m[0], (stalled || refused) ? 0xffffffffu : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_store(&m[1], crcv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			} else {
				m[0] = (stalled || refused) ? 0xffffffffu : 0u;
				m[1] = crcv;
			}
			a.split_ovf[b] = 0;
		}
		((uint8_t *)&m[2])[q & 3] = stalled ? 1 : 0;       // (a.wg_split > 1: the emit kernel looks at every sharer's byte)
		if (stalled && a.stalls)
			atomicAdd(a.stalls, 1u);
		if (beside) {
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__hip_atomic_exchange(&a.ready[32 * (size_t)(a.span_sub ? b : bi)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
}

// the gate in front of a parse that runs BESIDE its emit kernel (launch_wg): one wavefront that waits until `want` emit
// wavefronts have said they are resident -- they take their slices of the CUs' LDS and registers first, the parse workgroups fit in
// what is left.  Bounded (~30 ms): a gate that gives up only costs the placement.
__global__ __launch_bounds__(64) void k_gate(uint32_t *arrived, uint32_t want, uint32_t max_spins, uint32_t *poison)
{
	for (uint32_t spins = 0; spins < max_spins; spins++) {
		if (__hip_atomic_fetch_add(arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want)
			return;
		__builtin_amdgcn_s_sleep(64);
	}
	// (SPAN: the gate in front of a parse that overwrites records -- giving up there must not go unnoticed: DeflateArgs::poison)
	if (poison && threadIdx.x == 0)
		__hip_atomic_store(poison, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

void launch_emit_wg(const DeflateArgs &s, hipStream_t st);      // hd_emit_wg.hpp

// BESIDE (round 5, its last hours; DESIGN.md 4.2c, profiles/r05_wg_beside.txt).  The parse holds a CU with ONE workgroup (131 KB of its
// 160 KB of LDS, sixteen wavefronts of 96 registers); the emit-only kernel is one wavefront per member with 9.6 KB of LDS and <= 128
// registers.  One behind the other, the emit kernel was a quarter of these levels' time on its own.  Now the emit kernel is launched
// FIRST, on a second stream of the low priority class, ONCE per launch: 1536 candidate wavefronts of which every CU keeps the ones in its
// three lowest LDS blocks, one per SIMD (what a parse workgroup leaves of a CU's LDS and of a SIMD's registers holds:
// tools/coresidency_probe.hip, coresidency_real.hip, beside_filter_probe.hip); a one-wavefront gate on the caller's stream waits until
// the candidates have decided; the parses follow, sub-batch by sub-batch.  Every parse workgroup writes its records THROUGH its XCD's
// L2 (sc1 stores: the reader sits behind another XCD's L2; a release fence instead writes the whole L2 back, a million times per
// sub-batch of BGZF blocks) and raises the block's flag (an RMW, a flag per 128-byte line) once its stores are acknowledged; emit
// wavefronts take blocks from a counter per sub-batch, wait for the block's flag with sparse RMW polls (never a cached copy), read the
// records with loads that are coherent at the device's level themselves (sc1) and write the member.  SPAN: the records of the
// sub-batches alternate between two buffers; the parse of sub-batch k + 2 follows a launch of the emit kernel for what is left of
// sub-batch k and a gate on its members.  Behind the last parse the same kernel runs once more at full occupancy for whatever has not
// been taken.  The bytes do not depend on any of this.  What it needed, found the hard way: the parse's LDS passed at LAUNCH --
// declared statically, the compiler pads the kernel's registers so that nothing else fits the SIMD -- and then at a CONSTANT offset
// of that allocation (HD_WG_LDS_AT_ZERO_BASE), or every LDS address costs an add.
// Gain: config 5 (1 MiB members, level 6) 117.6 -> 131.8 GB/s, BGZF-sized blocks at level 6 115.6 -> 125.1, level 3 155 -> 175; the parse
// runs a fifth slower beside the emit wavefronts, the emit kernel's own time is gone.
constexpr size_t WG_LDS_DYNAMIC = sizeof(WgLds) + (HD_WG_LDS_AT_ZERO ? HD_WG_LDS_AT_ZERO_BASE : 0);     // what a BESIDE parse is launched with
struct WgBeside {
	hipStream_t side = nullptr;
	hipEvent_t ready = nullptr, done = nullptr;
	int init()
	{
		if (side)
			return 0;
		// A stream of its OWN priority class: the runtime keeps a pool of hardware queues per priority, and streams of one class
		// share queues once there are more of them than queues (four by default) -- two kernels in one hardware queue run one
		// behind the other, and this scheme NEEDS its two kernels to run at the same time (seen in the GPU suite, with the pipes'
		// and contexts' streams of earlier tests alive: the parse queued behind the emit kernel that was waiting for it, 2 s per
		// wait, members stored).  The callers' streams and every stream this library makes are of the default class.
		int lo = 0, hi = 0;
		if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess || lo == hi ||
		    hipStreamCreateWithPriority(&side, hipStreamNonBlocking, lo) != hipSuccess) {
			side = nullptr;
			return -1;                                     // (no second class of queues: the emit kernel follows the parse as ever)
		}
		if (hipEventCreateWithFlags(&ready, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&done, hipEventDisableTiming) != hipSuccess) {
			release();
			return -1;
		}
		return 0;
	}
	void release()
	{
		if (ready) (void)hipEventDestroy(ready);
		if (done) (void)hipEventDestroy(done);
		if (side) (void)hipStreamDestroy(side);
		ready = done = nullptr;
		side = nullptr;
	}
};
#ifndef HD_BESIDE_PARSE_PRIO
#define HD_BESIDE_PARSE_PRIO 0                   // experiment switches of tools/r05_prio.sh (issue priority of the two kernels' wavefronts, emit wavefronts kept per CU)
#endif
constexpr uint32_t WG_BESIDE_WAVES = 768;        // three per CU ...
constexpr uint32_t WG_BESIDE_CANDIDATES = 1536;  // ... kept from this many that are launched (k_deflate_dynamic<..., BESIDE>: a CU keeps those in its three lowest LDS blocks)
constexpr uint32_t WG_BESIDE_MIN = 512;          // blocks in a sub-batch below which the emit kernel simply follows the parse
constexpr uint32_t WG_BESIDE_MAX_BLOCK = 2097152; // ... and the longest block it is used for (one wavefront writes a member: 5 ms per MiB is the launch's tail)

// blocks [first, first + count) of a sub-batch: the workgroup parse, then the emit-only kernel over its records
inline void launch_wg(const DeflateArgs &a, int level, hipStream_t st)
{
	{
		// (more than 64 KB of dynamic LDS has to be asked for, once per device and kernel)
		static bool asked[64];
		int dev = 0;
		(void)hipGetDevice(&dev);
		if (dev >= 0 && dev < 64 && !asked[dev]) {
			(void)hipFuncSetAttribute((const void *)k_parse_wg<4, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WG_LDS_DYNAMIC);
			(void)hipFuncSetAttribute((const void *)k_parse_wg<2, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WG_LDS_DYNAMIC);
			(void)hipFuncSetAttribute((const void *)k_parse_wg<1, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WG_LDS_DYNAMIC);
			(void)hipFuncSetAttribute((const void *)k_parse_wg<1, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WG_LDS_DYNAMIC);
			asked[dev] = true;
		}
	}
	const uint32_t sub = wg_sub_batch(a.nblocks, a.split_max);
	DeflateArgs s = a;
	// scratch: [ overflow flags, one u32 per block (always 0: nothing stands behind this path) | records of one sub-batch ]
	s.split_ovf = (uint32_t *)a.scratch;
	s.scratch = a.scratch + (((uint64_t)a.nblocks * 4 + 15) & ~(uint64_t)15);
	s.wg = 1;
	uint8_t *const records = s.scratch;
	bool beside_off = false;                                 // (its set-up failed once: the launch goes on in the old order)
	for (uint32_t first = 0; first < a.nblocks; first += sub) {
		s.first = first;
		s.count = a.nblocks - first < sub ? a.nblocks - first : sub;
		// a handful of blocks wanted back soon (a.lat: each at most 64 KiB, the ring holds it whole): four or two workgroups
		// share a block's parse while that leaves the 256 CUs room for all of them
		s.wg_split = !a.lat ? 1u : s.count <= 64 ? 4u : s.count <= 128 ? 2u : 1u;
		s.stage_in = nullptr;
		if (s.wg_split > 1) {
			// (the staging area lies behind the records: wg_scratch_bytes(..., lat))
			uint8_t *stage = s.scratch + (uint64_t)sub * wg_layout(a.split_max).bytes;
			stage = (uint8_t *)(((uintptr_t)stage + 255) & ~(uintptr_t)255);
			hipLaunchKernelGGL(k_stage_in, dim3(s.count), dim3(1024), 0, st, s, stage);
			s.stage_in = stage;
		}
		WgBeside *bs = (!beside_off && !a.lat && a.nblocks >= WG_BESIDE_MIN && a.split_max <= WG_BESIDE_MAX_BLOCK) ? (WgBeside *)a.beside : nullptr;
		s.ready = s.arrived = s.next = s.emitted = s.poison = nullptr;
		s.span_sub = 0;
		s.scratch_b = nullptr;
		const uint32_t k = first / sub;                          // the sub-batch
		if (bs) {
			// behind the first records (wg_beside_bytes): [ a second buffer of records, when the launch is more than one sub-batch: SPAN ]
			// [ a flag line per block of the launch (SPAN) or of the sub-batch ] [ a counter per sub-batch ] [ arrival, hand-out ]
			const bool span = wg_beside_span(a.nblocks, a.split_max);
			const uint64_t rec_all = (uint64_t)sub * wg_layout(a.split_max).bytes;
			uint8_t *const rec_a = records;
			uint8_t *const rec_b = span ? (uint8_t *)(((uintptr_t)(rec_a + rec_all) + 255) & ~(uintptr_t)255) : nullptr;
			uint32_t *const flags = (uint32_t *)(((uintptr_t)((span ? rec_b : rec_a) + rec_all) + 255) & ~(uintptr_t)255);
			const uint64_t flagged = span ? a.nblocks : sub;
			uint32_t *const emitted = flags + 32 * (flagged + 1);
			const size_t per_sub = ((size_t)(a.nblocks / sub + 2) + 63) & ~(size_t)63;
			uint32_t *const taken = emitted + per_sub;               // SPAN: the hand-out counter of every sub-batch
			uint32_t *const counters = taken + per_sub;
			const uint32_t eg = WG_BESIDE_CANDIDATES;
			s.ready = flags;
			s.arrived = counters;
			s.next = counters + 16;
			s.poison = counters + 32;
			if (span) {
				s.span_sub = sub;
				s.scratch_b = rec_b;
				s.emitted = emitted;
				s.next = taken;
				s.scratch = (k & 1) ? rec_b : rec_a;
			}
			if (!span || k == 0) {
				// the emit wavefronts of this sub-batch -- SPAN: of the whole launch -- go first, on the side stream
				DeflateArgs e = s;
				if (span) {
					e.first = 0;
					e.count = a.nblocks;
					e.scratch = rec_a;
				}
				const size_t zero = (size_t)((uint8_t *)(counters + WG_BESIDE_COUNTER_WORDS) - (uint8_t *)flags);
				if (hipMemsetAsync(flags, 0, zero, st) != hipSuccess || hipEventRecord(bs->ready, st) != hipSuccess ||
				    hipStreamWaitEvent(bs->side, bs->ready, 0) != hipSuccess) {
					bs = nullptr;
					beside_off = true;
					s.ready = s.arrived = s.next = s.emitted = s.poison = nullptr;
					s.span_sub = 0;
					s.scratch = records;
				} else {
					hipLaunchKernelGGL((k_deflate_dynamic<HD_L2_WIN_BITS, HD_L2_HASH_BITS, HD_L2_MIN_LEN, 0, 1, 0, 0, 0, 1>), dim3(eg), dim3(64), 0, bs->side, e);
					(void)hipEventRecord(bs->done, bs->side);
					hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, st, s.arrived, eg, 1u << 14, (uint32_t *)nullptr);
				}
			} else if (k >= 2) {
				// SPAN: this parse overwrites the records of sub-batch k - 2: behind a gate on that sub-batch's members.  The resident
				// wavefronts have normally written them long ago; but how many of them there are is the dispatcher's business (none, with
				// another process's residents in the CUs' low LDS), so in front of the gate the emit kernel is launched for what is LEFT of
				// sub-batch k - 2 -- at full occupancy, that sub-batch's hand-out counter only: nothing to do as a rule, the old order at worst --
				// and the gate then waits for takers that are all alive.  If it gives up all the same, the parses behind it are told (poison)
				const uint32_t before = (k - 2) * sub, cnt = a.nblocks - before < sub ? a.nblocks - before : sub;
				DeflateArgs h = s;
				h.arrived = nullptr;
				h.first = 0;
				h.count = a.nblocks;
				h.scratch = rec_a;
				h.take_sub = k - 1;                          // (sub-batch k - 2, plus one)
				hipLaunchKernelGGL((k_deflate_dynamic<HD_L2_WIN_BITS, HD_L2_HASH_BITS, HD_L2_MIN_LEN, 0, 1, 0, 0, 0, 1>), dim3(cnt < 256u * 16u ? cnt : 256u * 16u), dim3(64), 0,
						   st, h);
				hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, st, emitted + (k - 2), cnt, 1u << 22, s.poison);
			}
		}
		const dim3 grid(s.count * s.wg_split), block(64 * HD_WG_WAVES);
		if (HD_WG_WAYS(level) == 4)
			{
			if (bs)
				hipLaunchKernelGGL((k_parse_wg<4, 1, 1>), grid, block, WG_LDS_DYNAMIC, st, s);
			else
				hipLaunchKernelGGL((k_parse_wg<4, 1>), grid, block, 0, st, s);
		}
		else if (HD_WG_WAYS(level) == 2)
			{
			if (bs)
				hipLaunchKernelGGL((k_parse_wg<2, 1, 1>), grid, block, WG_LDS_DYNAMIC, st, s);
			else
				hipLaunchKernelGGL((k_parse_wg<2, 1>), grid, block, 0, st, s);
		}
		else if (HD_WG_LAZY(level))
			{
			if (bs)
				hipLaunchKernelGGL((k_parse_wg<1, 1, 1>), grid, block, WG_LDS_DYNAMIC, st, s);
			else
				hipLaunchKernelGGL((k_parse_wg<1, 1>), grid, block, 0, st, s);
		}
		else
			{
			if (bs)
				hipLaunchKernelGGL((k_parse_wg<1, 0, 1>), grid, block, WG_LDS_DYNAMIC, st, s);
			else
				hipLaunchKernelGGL((k_parse_wg<1, 0>), grid, block, 0, st, s);
		}
		if (a.lat) {
			// the per-block boundary (HD_FRAME_LATENCY, blocks up to 64 KiB): the member written by a workgroup, the same bytes
			launch_emit_wg(s, st);
			continue;
		}
		if (bs) {
			if (s.span_sub && first + sub < a.nblocks) {
				continue;                                    // SPAN: the next sub-batch's parse follows at once, into the other buffer
			}
			// ... and behind the (last) parse the same kernel once more, at full occupancy, for what the resident wavefronts have not
			// taken yet (every flag is up by then): where the emit work outweighs the parse (level 3) they are the tail
			DeflateArgs h = s;
			h.arrived = nullptr;
			if (s.span_sub) {
				h.first = 0;
				h.count = a.nblocks;
				h.scratch = records;
			}
			const uint32_t hn = s.span_sub ? a.nblocks : s.count;
			const uint32_t hg = hn < 256u * 16u ? hn : 256u * 16u;
			hipLaunchKernelGGL((k_deflate_dynamic<HD_L2_WIN_BITS, HD_L2_HASH_BITS, HD_L2_MIN_LEN, 0, 1, 0, 0, 0, 1>), dim3(hg), dim3(64), 0, st, h);
			(void)hipStreamWaitEvent(st, bs->done, 0);           // the caller's stream carries on behind the members
			continue;
		}
		const uint32_t eg = s.count < 256u * 16u ? s.count : 256u * 16u;
		hipLaunchKernelGGL((k_deflate_dynamic<HD_L2_WIN_BITS, HD_L2_HASH_BITS, HD_L2_MIN_LEN, 0, 1>), dim3(eg), dim3(64), 0, st, s);
	}
}

} // namespace hd
