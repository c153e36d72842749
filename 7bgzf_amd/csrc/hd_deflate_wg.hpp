// hd_deflate_wg.hpp -- levels >= HD_WG_LEVEL, throughput form: the WORKGROUP parse (BASELINE config 5, "level-6-like").
//
// Replaces, for BGZF_METHOD=hip6..9, the matchfinder and the parser of libdeflate's lazy levels -- hc_matchfinder
// (lib/libdeflate/hc_matchfinder.h:183-338: hash chains of depth 35 over a 32 KiB window), deflate_compress_lazy_generic
// (lib/libdeflate/deflate_compress.c:2606-2809) and the block-split test (:2141-2218) -- with what ONE WORKGROUP holds in
// a CU's LDS: the whole DEFLATE window (a 64 KiB ring, distances up to 32768) and HD_WG_BUCKETS x HD_WG_WAYS table
// entries, shared by HD_WG_WAVES wavefronts (include/hipdeflate_params.h "WORKGROUP LEVELS" states the algorithm;
// oracle/hd_deflate_twin.c deflate_wg() is its serial statement and must give the same bytes).  Rounds 1-3 gave every
// wavefront a private ring and table: 8 KiB + 2560 two-way buckets was what a share of LDS held, and "level 6" came out
// at libdeflate-1's ratio.
//
// The 64-position steps of a block are dealt to the wavefronts round robin (wavefront w takes steps w, w + NW, ...); a
// step is one wavefront's from its first instruction to its last, so nothing but the ring, the table and a few words of
// hand-over state lives in LDS.  Two things have an order, and each is a TURN counter the wavefronts pass on:
//   A  the table: a step's lanes must read their buckets as the steps before left them.  In its turn a wavefront reads
//      its 64 buckets (one ds_read_b64 per lane), stores { itself, the three newest before } and hands the turn on;
//   C  the parse: which lanes start a token depends on where the last match of the steps before ended.  Outside its
//      turn a wavefront has verified its candidates (four of the bucket + the run candidate, 16 bytes each), extended the
//      full-span ones, applied the lazy rule and walked the parse of its step FROM LANE 0 (literal runs by a 64-bit carry
//      chain, one scalar hop per match).  In its turn it only merges: from the true entry lane it walks until it meets a
//      start of that canonical parse -- parses that share a start are equal from there on -- and hands on the next entry,
//      the token count and the split statistics.
// Everything else of a step -- hashing, 30 ring dwords per lane, the token words, the slab stores, the histogram
// atomics -- runs beside other wavefronts' turns.  Wavefront 0 also keeps the ring filled (one 1 KiB piece per step of
// its own, 8 KiB ahead) and folds the block's CRC-32 from the pieces as they pass.
//
// Output: the split path's records (hd_deflate_static.hpp SplitLayout) -- tokens, one histogram per DEFLATE block, CRC --
// which the emit-only kernel of hd_deflate_dynamic.hpp turns into members.  A block of any length is one stream.
#pragma once
#include "hd_deflate_dynamic.hpp"

namespace hd {

constexpr uint32_t WG_NW = HD_WG_WAVES;
constexpr uint32_t WG_AHEAD = 8;               // pieces the ring is filled ahead of wavefront 0's step
constexpr uint32_t WG_SPIN_LIMIT = 1u << 22;   // a turn that does not come: the block is given up (stored), never a hang

struct WgLds {
	__attribute__((aligned(16))) uint32_t ring32[HD_WG_RING / 4 + 8];      // + 32 bytes that mirror the start: unaligned reads never wrap
	__attribute__((aligned(16))) uint32_t table[HD_WG_BUCKETS * HD_WG_WAYS / 2];   // (p + 1) mod 2^16 x 4 per bucket, newest first
	uint32_t hist[2][320];                 // litlen [0,288), offset [288,320): DEFLATE block d counts in hist[d & 1]
	// The parse state, owned by whoever holds the C turn -- ONE record that a wavefront reads with one ds_read_b32 (lane i
	// word i): the spin for the turn and the fetch of the state are the same instruction.  LDS executes a wavefront's
	// instructions in order and one at a time, so a reader that sees the turn word sees everything stored before it.
	uint32_t st[16];
	uint32_t a_turn;                       // the step whose table access may run
	uint32_t flushed_db;                   // DEFLATE blocks whose histogram has left for HBM
	uint32_t done[WG_NW];                  // steps wavefront w has finished
};
// words of WgLds::st
enum { WG_TURN = 0, WG_E, WG_NTOK, WG_NOBS0, WG_NOBS1, WG_NOBS2, WG_DB, WG_DBTOK0, WG_BLKBEGIN, WG_OBS0, WG_OBS1, WG_OBS2, WG_NMERGED,
       WG_FAIL };
#define WG_BARRIER() asm volatile("" ::: "memory")
// LDS words that other wavefronts write are read and written through address-space-3 pointers (ds_read / ds_write, never
// flat_*: the hand-over argument above is about ONE queue, the LDS unit's)
#define WG_LDS __attribute__((address_space(3)))
typedef volatile WG_LDS uint32_t *wg_word_p;

// wait until *word >= want (words only grow); false if the workgroup has failed or the word does not come
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
		__builtin_amdgcn_s_sleep(1);
	}
}

// 16 bytes at ring offset `o` (any alignment): five dwords, funnel-shifted
__device__ __forceinline__ uint4 wg_read16(const uint32_t *ring32, uint32_t o)
{
	const uint32_t *p = ring32 + (o >> 2);
	const uint32_t d0 = p[0], d1 = p[1], d2 = p[2], d3 = p[3], d4 = p[4];
	const uint32_t sh = o & 3;
	return make_uint4(__builtin_amdgcn_alignbyte(d1, d0, sh), __builtin_amdgcn_alignbyte(d2, d1, sh),
			  __builtin_amdgcn_alignbyte(d3, d2, sh), __builtin_amdgcn_alignbyte(d4, d3, sh));
}
__device__ __forceinline__ uint64_t wg_read8(const uint32_t *ring32, uint32_t o)
{
	const uint32_t *p = ring32 + (o >> 2);
	const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
	const uint32_t sh = o & 3;
	return (uint64_t)__builtin_amdgcn_alignbyte(d1, d0, sh) | ((uint64_t)__builtin_amdgcn_alignbyte(d2, d1, sh) << 32);
}
// bytes that a and b have in common from the front, 0..16
__device__ __forceinline__ uint32_t wg_common16(uint4 a, uint4 b)
{
	const uint64_t x0 = ((uint64_t)(a.y ^ b.y) << 32) | (a.x ^ b.x), x1 = ((uint64_t)(a.w ^ b.w) << 32) | (a.z ^ b.z);
	const uint32_t m0 = x0 ? (uint32_t)__builtin_ctzll(x0) >> 3 : 8u;
	const uint32_t m1 = x1 ? (uint32_t)__builtin_ctzll(x1) >> 3 : 8u;
	return m0 < 8 ? m0 : 8 + m1;
}

// The parse of a step from lane `b0` (scalar code: every value is uniform).  lit = lanes whose token would be a literal,
// flen = per lane the length of its match (lanes of ~lit), lanes = positions of the step.  Literal runs are one 64-bit
// addition (the carry ripples through the run and lands on the match lane behind it), a match is one v_readlane.
// With stop != 0 the walk ends where it meets a lane of `stop` (a start of the parse from lane 0: from a common start on
// two parses are the same): merged = true, *at = that lane.  Returns the starts it has marked; *exit_lane = where the walk
// left the step (>= lanes) when it did not merge.
__device__ __forceinline__ uint64_t wg_walk(uint32_t b0, uint64_t lit, uint32_t flen, uint32_t lanes, uint64_t stop, bool &merged,
					    uint32_t &at, uint32_t &exit_lane)
{
	uint64_t R = 0;
	uint32_t b = b0;
	merged = false;
	at = 0;
	while (b < lanes) {
		const uint64_t x = 1ull << b;
		if (lit & x) {
			const uint64_t t = lit + x;
			const bool over = t < x;                         // the run reaches lane 63
			const uint64_t run = t ^ lit;                    // lanes b .. k, k = the first lane behind the run
			const uint64_t hit = stop & run;
			if (hit) {
				at = (uint32_t)__builtin_ctzll(hit);
				R |= run & ((1ull << at) - 1);
				merged = true;
				break;
			}
			R |= run;
			if (over) {
				b = 64;
				break;
			}
			b = 63 - (uint32_t)__builtin_clzll(run);
			if (b >= lanes)                                  // (lit holds no lane >= lanes: the run ended at the step's end)
				break;
		} else {
			if (stop & x) {
				at = b;
				merged = true;
				break;
			}
			R |= x;
		}
		b += readlane(flen, b);                              // lane b is a match start
	}
	exit_lane = b;
	return R;
}

__global__ __launch_bounds__(64 * HD_WG_WAVES) void k_parse_wg(DeflateArgs a)
{
	__shared__ WgLds L;
	const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const uint32_t bi = blockIdx.x, b = a.first + bi;
	const uint8_t *src = a.in + a.in_off[b];
	const uint32_t n = a.in_len[b];
	const bool aligned = (((uintptr_t)src) & 15) == 0;
	const CrcTables *ct = a.ct;
	const SplitLayout lay = wg_layout(a.split_max);
	uint8_t *const rec = a.scratch + (uint64_t)bi * lay.bytes;
	uint32_t *const tok = (uint32_t *)rec;
	uint32_t *const rec_ntok = (uint32_t *)(rec + lay.off_ntok);
	uint32_t *const rec_hist = (uint32_t *)(rec + lay.off_hist);
	WG_LDS WgLds *const Lp = (WG_LDS WgLds *)&L;
	const wg_word_p vst = (wg_word_p)Lp->st, vfail = vst + WG_FAIL, va_turn = (wg_word_p)&Lp->a_turn,
			vflushed = (wg_word_p)&Lp->flushed_db, vdone = (wg_word_p)Lp->done;

	// ---- LDS: table and histograms zero, state words, the first pieces of the block ---------------------------------
	for (uint32_t i = threadIdx.x; i < HD_WG_BUCKETS * HD_WG_WAYS / 8; i += 64 * WG_NW)
		((uint4 *)L.table)[i] = make_uint4(0, 0, 0, 0);
	for (uint32_t i = threadIdx.x; i < 640; i += 64 * WG_NW)
		(&L.hist[0][0])[i] = 0;
	if (threadIdx.x < WG_NW)
		L.done[threadIdx.x] = 0;
	if (threadIdx.x < 16)
		L.st[threadIdx.x] = 0;
	if (threadIdx.x == 0)
		L.a_turn = L.flushed_db = 0;
	CrcLanes crc;
	uint32_t next_piece = 0;                   // wavefront 0: the next piece to go into the ring
	uint4 pend = make_uint4(0, 0, 0, 0);       // ... and its bytes, requested a step of wavefront 0 earlier
	const uint32_t npieces = (n + HD_PIECE - 1) / HD_PIECE;
	auto put_piece = [&](uint32_t piece, uint4 v) {
		const uint32_t o = (piece * HD_PIECE + 16 * lane) & (HD_WG_RING - 1);
		*(uint4 *)((uint8_t *)L.ring32 + o) = v;
		if (o < 32)                        // the mirror behind the ring's end
			*(uint4 *)((uint8_t *)L.ring32 + HD_WG_RING + o) = v;
		crc.fold(ct, piece, piece * HD_PIECE + 16 * lane + 16 <= n, v);
	};
	if (w == 0) {
		crc.init(lane, n);
		for (; next_piece < npieces && next_piece <= WG_AHEAD; next_piece++)
			put_piece(next_piece, load_slot(src, n, next_piece, lane, aligned));
		if (next_piece < npieces)
			pend = load_slot(src, n, next_piece, lane, aligned);
	}
	__syncthreads();

	HashConsts6 hk;
	hk.init(2 * HD_WG_BUCKETS);                // byte offset of an 8-byte bucket: 4 * (2 * slot)
	hk.m = 0xfff8u;
	const uint32_t nsteps = (n + 63) / 64;
	uint32_t my_done = 0;
	WG_LDS uint8_t *const tab8 = (WG_LDS uint8_t *)Lp->table;

	for (uint32_t s = w; s < nsteps; s += WG_NW) {
		const uint32_t S = s * 64, p = S + lane;
		const uint32_t lanes = n - S < 64 ? n - S : 64;
		const uint64_t lanem = lanes == 64 ? ~0ull : (1ull << lanes) - 1;
		if (w == 0 && s) {
			// the ring moves on: the piece requested a round ago goes in (its slot's old bytes are 55 KiB behind every
			// reader), the next one is requested
			if (next_piece < npieces) {
				put_piece(next_piece, pend);
				next_piece++;
				if (next_piece < npieces)
					pend = load_slot(src, n, next_piece, lane, aligned);
			}
		}
		// ---- own bytes, hash ----------------------------------------------------------------------------------
		const uint4 own = wg_read16(L.ring32, p & (HD_WG_RING - 1));
		const bool keyed = p + HD_LAZY_KEY_BYTES <= n;
		const uint32_t haddr = hash_slot_addr6(own.x, own.y, hk);
		// ---- A: the table, in step order ---------------------------------------------------------------------
		if (!wg_wait(va_turn, s, vfail))
			break;
		uint2 old = make_uint2(0, 0);
		if (keyed) {
			const unsigned long long o64 = *(WG_LDS const unsigned long long *)(tab8 + haddr);        // ds_read_b64
			old = make_uint2((uint32_t)o64, (uint32_t)(o64 >> 32));
			// two dword stores, not one of eight bytes: of the lanes of a step that share a bucket the HIGHEST keeps
			// each store (the LDS-order probe of ctx_init checks ds_write_b32), so both halves are that lane's
			*(wg_word_p)(tab8 + haddr) = (old.x << 16) | ((p + 1) & 0xffffu);
			*(wg_word_p)(tab8 + haddr + 4) = (old.y << 16) | (old.x >> 16);
		}
		WG_BARRIER();
		if (lane == 0)
			*va_turn = s + 1;
		WG_BARRIER();
		// ---- verify: the byte before (inside the step), then the bucket newest first -------------------------
		const uint32_t room = n - p < HD_WG_VCAP ? (p < n ? n - p : 0u) : HD_WG_VCAP;
		uint32_t best = 0, dist = 0;
#pragma unroll
		for (int k = 0; k <= HD_WG_WAYS; k++) {
			uint32_t back;
			bool ok;
			if (k == 0) {
				back = 1;
				ok = keyed && lane != 0;
			} else {
				const uint32_t e = k == 1 ? (old.x & 0xffffu) : k == 2 ? (old.x >> 16) : k == 3 ? (old.y & 0xffffu) : (old.y >> 16);
				back = (p + 1 - e) & 0xffffu;
				ok = keyed && e != 0 && back != 0 && back <= HD_WG_WINDOW && back <= p;
			}
			const uint4 c = wg_read16(L.ring32, (p - back) & (HD_WG_RING - 1));
			uint32_t m = wg_common16(own, c);
			m = m < room ? m : room;
			if (ok && m > best) {
				best = m;
				dist = back;
			}
		}
		const bool cand = best >= HD_WG_MIN_LEN;
		const uint32_t clen = cand ? best : 0u;
		// ---- a match of the whole verified span is extended to its full length -------------------------------
		uint32_t flen = clen;
		{
			const uint32_t maxlen = n - p < HD_MAX_MATCH ? (p < n ? n - p : 0u) : HD_MAX_MATCH;
			bool act = cand && best == HD_WG_VCAP && flen < maxlen;
			while (__ballot(act)) {
				const uint64_t x = wg_read8(L.ring32, (p + flen) & (HD_WG_RING - 1)) ^
						   wg_read8(L.ring32, (p + flen - dist) & (HD_WG_RING - 1));
				uint32_t adv = x ? (uint32_t)__builtin_ctzll(x) >> 3 : 8u;
				const bool more = adv == 8;
				adv = adv < maxlen - flen ? adv : maxlen - flen;
				if (act)
					flen += adv;
				act = act && more && flen < maxlen;
			}
		}
		// ---- the lazy rule on the lane to the right (deflate_compress.c:2723-2726, lengths capped at 16) ------
		const uint32_t clen_r = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)clen, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
		const uint32_t dist_r = (uint32_t)__builtin_amdgcn_update_dpp(1, (int)dist, 0x130, 0xf, 0xf, false);
		const int gain = 4 * ((int)clen_r - (int)clen) + ((int)(31 - __clz(dist | 1)) - (int)(31 - __clz(dist_r | 1)));
		const bool defer = cand && lane + 1 < lanes && clen_r != 0 && clen_r >= clen && gain > 2;
		const uint64_t take = __ballot(cand && !defer) & lanem;
		const uint64_t lit = ~take & lanem;
		const uint64_t long9 = __ballot(flen >= 9);
		// ---- the parse of this step from lane 0 -----------------------------------------------------------------
		bool mg;
		uint32_t at, exit0;
		const uint64_t R0 = wg_walk(0, lit, flen, lanes, 0, mg, at, exit0);
		// token words and symbols (for the lanes that turn out to be starts)
		const bool is_take = (take >> lane) & 1;
		const uint32_t tw = is_take ? (HD_TOKEN_MATCH | ((flen - 3) << 16) | (dist - 1)) : (own.x & 0xffu);
		uint32_t lsym = own.x & 0xffu, dsym = 0;
		if (is_take) {
			uint32_t eb, ev;
			len_slot(flen, lsym, eb, ev);
			lsym += 257;
			off_slot(dist, dsym, eb, ev);
		}
		// ---- C: the merge, in step order ----------------------------------------------------------------------------
		uint32_t rec_v;
		{
			bool okc = true;
			for (uint32_t spins = 0;; spins++) {
				rec_v = vst[lane & 15];
				WG_BARRIER();
				if (readlane(rec_v, WG_TURN) == s)
					break;
				if (readlane(rec_v, WG_FAIL) || spins > WG_SPIN_LIMIT) {
					*vfail = 1;
					okc = false;
					break;
				}
				__builtin_amdgcn_s_sleep(1);
			}
			if (!okc)
				break;
		}
		const uint32_t E = readlane(rec_v, WG_E), tok0 = readlane(rec_v, WG_NTOK), db = readlane(rec_v, WG_DB);
		uint64_t starts = 0;
		uint32_t Enew = E;
		if (E < S + lanes) {
			const uint32_t e = E > S ? E - S : 0u;
			uint32_t ex;
			starts = wg_walk(e, lit, flen, lanes, R0, mg, at, ex);
			if (mg) {
				starts |= R0 & ~((1ull << at) - 1);
				ex = exit0;
			}
			starts &= lanem;
			Enew = S + ex;
		}
		const uint32_t ntk = (uint32_t)__popcll(starts);
		const uint32_t c_lit = (uint32_t)__popcll(starts & lit), c_long = (uint32_t)__popcll(starts & take & long9);
		const uint32_t c_short = ntk - c_lit - c_long;
		// the open DEFLATE block ends behind this step? (never behind the last one)
		bool close = false, merge_obs = false;
		const uint32_t here = S + lanes;
		uint32_t o0 = readlane(rec_v, WG_NOBS0) + c_lit, o1 = readlane(rec_v, WG_NOBS1) + c_short, o2 = readlane(rec_v, WG_NOBS2) + c_long;
		const uint32_t db_tok0 = readlane(rec_v, WG_DBTOK0), blk_begin = readlane(rec_v, WG_BLKBEGIN);
		uint32_t m0 = 0, m1 = 0, m2 = 0, nm = 0;
		if (here < n) {
			close = tok0 + ntk - db_tok0 >= HD_DYN_BLOCK_TOKENS;
			const uint32_t nn = o0 + o1 + o2;
			if (!close && nn >= HD_WG_SPLIT_OBS && here - blk_begin >= HD_WG_SPLIT_MIN && n - here >= HD_WG_SPLIT_MIN) {
				// the observation test (twin: wg_split_check)
				m0 = readlane(rec_v, WG_OBS0);
				m1 = readlane(rec_v, WG_OBS1);
				m2 = readlane(rec_v, WG_OBS2);
				nm = readlane(rec_v, WG_NMERGED);
				if (nm > 0) {
					const uint32_t e0 = m0 * nn, a0 = o0 * nm, e1 = m1 * nn, a1 = o1 * nm, e2 = m2 * nn, a2 = o2 * nm;
					const uint32_t total = (a0 > e0 ? a0 - e0 : e0 - a0) + (a1 > e1 ? a1 - e1 : e1 - a1) + (a2 > e2 ? a2 - e2 : e2 - a2);
					const uint32_t items = nm + nn, blen = here - blk_begin;
					uint32_t cutoff = nn * 200u / 512u * nm;
					if (blen < 10000 && items < 8192)
						cutoff += (cutoff >> 13) * (8192u - items);
					close = total + (blen / 4096u) * nm >= cutoff;
				}
				if (!close) {
					merge_obs = true;
					m0 += o0;
					m1 += o1;
					m2 += o2;
					nm += nn;
					o0 = o1 = o2 = 0;
				}
			}
		}
		{
			// the new state: lane i word i (what has not changed is not stored)
			const uint32_t ntok1 = tok0 + ntk;
			uint32_t nv = lane == WG_E ? Enew : lane == WG_NTOK ? ntok1 : lane == WG_NOBS0 ? o0 : lane == WG_NOBS1 ? o1 : o2;
			uint64_t wm = (1ull << WG_E) | (1ull << WG_NTOK) | (1ull << WG_NOBS0) | (1ull << WG_NOBS1) | (1ull << WG_NOBS2);
			if (close) {
				nv = lane == WG_E ? Enew : lane == WG_NTOK ? ntok1 : lane == WG_DB ? db + 1 : lane == WG_DBTOK0 ? ntok1 : lane == WG_BLKBEGIN ? here : 0u;
				wm = 0x1ffeull;                                      // words 1 .. 12
			} else if (merge_obs) {
				nv = lane == WG_OBS0 ? m0 : lane == WG_OBS1 ? m1 : lane == WG_OBS2 ? m2 : lane == WG_NMERGED ? nm : nv;
				wm |= (1ull << WG_OBS0) | (1ull << WG_OBS1) | (1ull << WG_OBS2) | (1ull << WG_NMERGED);
			}
			if ((wm >> lane) & 1)
				vst[lane] = nv;
			WG_BARRIER();
			if (lane == 0)
				vst[WG_TURN] = s + 1;
			WG_BARRIER();
		}
		// ---- the step's tokens: slab, histogram of the open DEFLATE block -------------------------------------------
		// hist[db & 1] was block db - 2's: its counts must have left for HBM (they have, long ago: a block is >= 78 steps)
		if (db >= 2 && !wg_wait(vflushed, db - 1, vfail))
			break;
		if ((starts >> lane) & 1) {
			const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(starts >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)starts, 0));
			tok[tok0 + rank] = tw;
			atomicAdd(&L.hist[db & 1][lsym], 1u);
			if (is_take)
				atomicAdd(&L.hist[db & 1][288 + dsym], 1u);
		}
		WG_BARRIER();
		my_done++;
		if (lane == 0)
			vdone[w] = my_done;
		WG_BARRIER();
		if (close) {
			// this wavefront closed DEFLATE block db: when every step up to this one is through, its histogram leaves
			bool okd = true;
			for (uint32_t spins = 0;; spins++) {
				const uint32_t need = lane < WG_NW && s >= lane ? (s - lane) / WG_NW + 1 : 0u;     // steps of wavefront `lane` up to s
				const uint32_t have = lane < WG_NW ? vdone[lane] : 0u;
				WG_BARRIER();
				if (!__ballot(have < need))
					break;
				if (uniform(*vfail) || spins > WG_SPIN_LIMIT) {
					*vfail = 1;
					okd = false;
					break;
				}
				__builtin_amdgcn_s_sleep(1);
			}
			if (!okd)
				break;
			// (the closers take their turns too: block db - 1's histogram has left before this one does)
			if (!wg_wait(vflushed, db, vfail))
				break;
			for (uint32_t i = lane; i < 320; i += 64) {
				const wg_word_p h = (wg_word_p)&Lp->hist[db & 1][i];
				rec_hist[db * 320 + i] = *h;
				*h = 0;
			}
			if (lane == 0)
				rec_ntok[db] = tok0 + ntk - db_tok0;
			WG_BARRIER();
			if (lane == 0)
				*vflushed = db + 1;
			WG_BARRIER();
		}
	}
	__syncthreads();
	// ---- the last DEFLATE block's histogram, the record, the CRC ----------------------------------------------------
	if (w == 0) {
		while (next_piece < npieces) {             // (a block shorter than the look-ahead: nothing left; else the pending piece)
			put_piece(next_piece, pend);
			next_piece++;
			if (next_piece < npieces)
				pend = load_slot(src, n, next_piece, lane, aligned);
		}
		const uint32_t crcv = crc.finish(ct, lane, n, src + (n & ~15u));
		const bool failed = uniform(*vfail) != 0;
		const uint32_t db = uniform(vst[WG_DB]), ntok = uniform(vst[WG_NTOK]), db_tok0 = uniform(vst[WG_DBTOK0]);
		for (uint32_t i = lane; i < 320; i += 64)
			rec_hist[db * 320 + i] = L.hist[db & 1][i];
		if (lane == 0) {
			rec_ntok[db] = ntok - db_tok0;
			uint32_t *m = (uint32_t *)(rec + lay.off_rec);
			m[0] = failed ? 0xffffffffu : db + 1;
			m[1] = crcv;
			a.split_ovf[b] = 0;
		}
	}
}

// blocks [first, first + count) of a sub-batch: the workgroup parse, then the emit-only kernel over its records
inline void launch_wg(const DeflateArgs &a, hipStream_t st)
{
	const uint32_t sub = wg_sub_batch(a.nblocks, a.split_max);
	DeflateArgs s = a;
	// scratch: [ overflow flags, one u32 per block (always 0: nothing stands behind this path) | records of one sub-batch ]
	s.split_ovf = (uint32_t *)a.scratch;
	s.scratch = a.scratch + (((uint64_t)a.nblocks * 4 + 15) & ~(uint64_t)15);
	s.wg = 1;
	for (uint32_t first = 0; first < a.nblocks; first += sub) {
		s.first = first;
		s.count = a.nblocks - first < sub ? a.nblocks - first : sub;
		hipLaunchKernelGGL(k_parse_wg, dim3(s.count), dim3(64 * HD_WG_WAVES), 0, st, s);
		const uint32_t eg = s.count < 256u * 16u ? s.count : 256u * 16u;
		hipLaunchKernelGGL((k_deflate_dynamic<HD_L2_WIN_BITS, HD_L2_HASH_BITS, HD_L2_MIN_LEN, 0, 1>), dim3(eg), dim3(64), 0, st, s);
	}
}

} // namespace hd
