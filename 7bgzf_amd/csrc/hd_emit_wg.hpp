// hd_emit_wg.hpp -- levels >= HD_WG_LEVEL behind the PER-BLOCK boundary (HD_FRAME_LATENCY: the LD_PRELOAD hook, hip_deflate,
// hip_deflate_flush, i.e. bgzf_compress.c:163-169, lib/zlibutil.c:179-192, applet/7bgzf.c:183-217): the member of ONE block
// written by a WORKGROUP of sixteen wavefronts instead of by one.
//
// The reference has one codec per level whatever calls it (deflate_compress.c:3951-3955).  Rounds 3-4 did not: a lone block
// was cut into 8160-byte segments and 2 KiB parts for one-wavefront parsers, and "hip6" through the hook wrote more bytes than
// the reference's level 1.  Now the per-block boundary runs the workgroup parse of hd_deflate_wg.hpp on the whole block (a lone
// workgroup parses 64 KiB in well under 100 us) and this kernel turns its records into the member -- THE SAME BYTES the
// one-wavefront emit kernel of the batch path writes (hd_deflate_dynamic.hpp, the a.wg branch; the CPU twin's deflate_wg()),
// so the latency form of these levels is no longer a different stream, only a different schedule:
//   * every wavefront walks the pieces' counts and cuts the DEFLATE blocks alike (libdeflate's observation test,
//     deflate_compress.c:2141-2218; a few hundred scalar instructions);
//   * per DEFLATE block: all sixteen count the symbols of their pieces into one histogram; wavefront 0 builds the litlen
//     code while wavefront 1 builds the offset code; wavefront 0 makes the code-length RLE, the precode and the header and
//     decides dynamic / static / give up, wavefront 1 fills the 512-entry litlen table meanwhile; every wavefront prices
//     its pieces (one table load per token), a prefix sum over the pieces' bits gives each piece its place, and the pieces
//     are coded side by side into ONE staging image of the member in LDS (ds_or at bit granularity: 64 KiB + frame);
//   * the image leaves as 16-byte stores of the whole workgroup.
// Blocks up to 64 KiB (EW_ROOM_MAX of room): every BGZF block, everything the per-block codecs batch.  Longer ones keep the
// one-wavefront emit kernel (same bytes, launch_wg decides by the slot size).
#pragma once
#include "hd_deflate_wg.hpp"

namespace hd {

constexpr uint32_t EW_NW = 16;                      // wavefronts
constexpr uint32_t EW_MAX_PIECES = 64;              // pieces of a block: lane = piece in the cut scan
constexpr uint32_t EW_STAGE_DW = 16640;             // 66,560 bytes: the longest header (20) + a payload below the stored size of
                                                    // 64 KiB (65,546) + flush suffix (5) + trailer (8), and the three dwords a
                                                    // 48-bit field may touch
constexpr uint32_t EW_BLOCK_MAX = 65536;            // the longest block: the host sets DeflateArgs::lat only where it knows that no
                                                    // block of the launch is longer (a latency context's input stride, or a slot
                                                    // of at most this many bytes: the parse refuses a block longer than its room)

#define EW_LDS __attribute__((address_space(3)))
constexpr uint32_t EW_SCRATCH_DW = 1536;            // dwords of the image lent to an offset-code construction (>= sizeof(HuffScratch))
constexpr uint32_t EW_SLOTS = 4;                    // DEFLATE blocks whose codes are built side by side (a wavefront each)
struct EwSlot {
	DynBuild build;                                 // the block's construction scratch and codes
	DynLds L;                                       // its histograms, the precode
	uint32_t lut[512];                              // litlen half of a token: codeword (+ extra bits) | bit count << 24
	uint32_t info[8];                               // 0: the block's bits  1: dynamic code  2: header bits  3: code-length items  4: hlit  5: hdist  6: hclen
};
struct EwLds {
	__attribute__((aligned(16))) uint32_t stage[EW_STAGE_DW];
	EwSlot slot[EW_SLOTS];
	uint32_t ph[EW_MAX_PIECES][160];                // per piece: 320 symbol counts, 16 bits each (a piece holds <= 1024 tokens):
	                                                // litlen symbol s at half s, offset symbol d at half 288 + d
	uint32_t piece_bits[EW_MAX_PIECES];
	uint32_t cuts[EW_MAX_PIECES + 2];               // [0] the number of DEFLATE blocks, [1 + i] the piece behind block i
	uint32_t lsym[64];                              // length - 3 -> litlen symbol - 257, a byte each
	uint32_t ctl[8];                                // 3: member bytes
};

// one field per lane (nbits <= 32, 0 = none), in lane order, ORed into the image at bit `bitpos`; returns the bits placed
__device__ __forceinline__ uint32_t ew_emit1(EW_LDS uint32_t *stage, uint32_t bitpos, uint32_t code, uint32_t nbits)
{
	const uint32_t incl = wave_incl_scan(nbits);
	if (nbits) {
		const uint32_t bp = bitpos + incl - nbits, sh = bp & 31, i = bp >> 5;
		__hip_atomic_fetch_or(&stage[i], code << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		if (sh + nbits > 32)
			__hip_atomic_fetch_or(&stage[i + 1], code >> (32 - sh), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	}
	return readlane(incl, 63);
}

// One wavefront, one DEFLATE block (its histograms are in S->L): the codes, the RLE of their lengths, the precode, the exact
// cost, dynamic or static, the litlen table -- flush_block of the emit-only kernel (hd_deflate_dynamic.hpp) statement by
// statement, everything but the bits; what the header needs later stays in the slot.
#ifdef HD_EMIT_STATS
#define EWB_T(k) do { if (threadIdx.x == 0) { const unsigned long long t_now = clock64(); atomicAdd(&g_emit_stats[16 + (k)], t_now - t_b); t_b = t_now; } } while (0)
#else
#define EWB_T(k) do { } while (0)
#endif
// the litlen code of the slot's block (which = 0: the slot's own construction scratch) or its offset code (which = 1: scratch
// lent by the caller -- the two constructions of a block run side by side on two wavefronts)
__device__ __forceinline__ void ew_build_code(EW_LDS EwSlot *S, uint32_t which, EW_LDS HuffScratch *scratch, uint32_t lane)
{
#ifdef HD_EMIT_STATS
	unsigned long long t_b = clock64();
#endif
	// (build_code_t with LDS-typed pointers: ds_read / ds_write instead of flat instructions)
	if (which == 0) {
		if (lane == 0)
			S->L.lf[256] += 1;                              // end of block
		build_code_t<const EW_LDS uint32_t *, EW_LDS uint32_t *, EW_LDS HuffScratch *>(S->L.lf, 288, HD_LITLEN_MAXBITS, S->build.lcode, scratch, lane, true);
		EWB_T(0);
	} else {
		build_code_t<const EW_LDS uint32_t *, EW_LDS uint32_t *, EW_LDS HuffScratch *>(S->L.df, 32, HD_OFFSET_MAXBITS, S->build.dcode, scratch, lane, true);
	}
}

__device__ __forceinline__ void ew_prepare_block(EW_LDS EwSlot *S, uint32_t lane)
{
#ifdef HD_EMIT_STATS
	unsigned long long t_b = clock64();
#endif
	EW_LDS uint32_t *const lf = S->L.lf, *const df = S->L.df, *const pfreq = S->L.pfreq, *const pcode = S->L.pcode;
	EW_LDS uint32_t *const lcode = S->build.lcode, *const dcode = S->build.dcode;
	EW_LDS uint16_t *const items = (EW_LDS uint16_t *)&S->build.hs.nf[64];           // (DynBuild::items() / lens())
	EW_LDS uint8_t *const lens = (EW_LDS uint8_t *)&S->build.hs.nf[64 + 160];
	EW_LDS uint16_t *const run_start = S->build.hs.parent;                           // free until the precode is built
	// (the litlen and offset codes are in place: ew_build_code, two wavefronts per block)
	if (lane < 19)
		pfreq[lane] = 0;
	uint32_t hlit, hdist;
	{
		const uint64_t ml = __ballot(lane < 29 && (lcode[257 + lane] >> 16) != 0);
		const uint64_t md = __ballot(lane < 29 && (dcode[1 + lane] >> 16) != 0);
		hlit = ml ? 257 + 64 - (uint32_t)__clzll((long long)ml) : 257;
		hdist = md ? 1 + 64 - (uint32_t)__clzll((long long)md) : 1;
	}
	const uint32_t total = hlit + hdist;
	for (uint32_t i = lane; i < total; i += 64)
		lens[i] = (uint8_t)((i < hlit ? lcode[i] : dcode[i - hlit]) >> 16);
	uint32_t nruns = 0;
	for (uint32_t base = 0; base < total; base += 64) {
		const uint32_t i = base + lane;
		const bool st = i < total && (i == 0 || lens[i] != lens[i - 1]);
		const uint64_t mm = __ballot(st);
		if (st)
			run_start[nruns + __popcll(mm & ((1ull << lane) - 1))] = (uint16_t)i;
		nruns += (uint32_t)__popcll(mm);
	}
	if (lane == 0)
		run_start[nruns] = (uint16_t)total;
	uint32_t ni = 0;
	for (uint32_t rb = 0; rb < nruns; rb += 64) {
		const uint32_t r = rb + lane;
		const bool valid = r < nruns;
		const uint32_t s0 = valid ? run_start[r] : 0, len = valid ? run_start[r + 1] - s0 : 0;
		const uint32_t v = valid ? lens[s0] : 0;
		uint32_t rep, big, rest, lead;
		if (v == 0) {
			rep = 18; lead = 0;
			big = len / 138; rest = len - 138 * big;
		} else {
			rep = 16; lead = valid ? 1u : 0u;
			big = (len - lead) / 6; rest = (len - lead) - 6 * big;
		}
		const uint32_t full = v == 0 ? 138u : 6u, base_len = v == 0 ? 11u : 3u;
		uint32_t extra_rep = 0, extra_sym = rep, extra_base = base_len;
		if (rest >= base_len) {
			extra_rep = 1;
		} else if (v == 0 && rest >= 3) {
			extra_rep = 1; extra_sym = 17; extra_base = 3;
		}
		const uint32_t tail = extra_rep ? 0u : rest;
		const uint32_t c = valid ? lead + big + extra_rep + tail : 0u;
		const uint32_t incl = wave_incl_scan(c);
		uint32_t o = ni + incl - c;
		if (valid) {
			if (lead)
				items[o++] = (uint16_t)v;
			for (uint32_t j = 0; j < big; j++)
				items[o++] = (uint16_t)(rep | ((full - base_len) << 8));
			if (extra_rep)
				items[o++] = (uint16_t)(extra_sym | ((rest - extra_base) << 8));
			for (uint32_t j = 0; j < tail; j++)
				items[o++] = (uint16_t)v;
			if (lead + tail)
				__hip_atomic_fetch_add(&pfreq[v], lead + tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			if (big)
				__hip_atomic_fetch_add(&pfreq[rep], big, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			if (extra_rep)
				__hip_atomic_fetch_add(&pfreq[extra_sym], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
		ni += readlane(incl, 63);
	}
	EWB_T(2);
	build_code_t<const EW_LDS uint32_t *, EW_LDS uint32_t *, EW_LDS HuffScratch *>(pfreq, 19, HD_PRECODE_MAXBITS, pcode, &S->build.hs, lane, true);
	EWB_T(3);
	uint32_t hclen = 19;
	while (hclen > 4 && (uniform(pcode[k_perm19[hclen - 1]]) >> 16) == 0)
		hclen--;
	uint32_t dyn = 0, sta = 0, extra = 0;
	for (uint32_t base = 0; base < ni; base += 64) {
		const uint32_t kx = base + lane;
		if (kx < ni) {
			const uint32_t sym = items[kx] & 31;
			dyn += (pcode[sym] >> 16) + (sym == 16 ? 2u : sym == 17 ? 3u : sym == 18 ? 7u : 0u);
		}
	}
	// (what the header of the dynamic code takes: the 17 + 3 hclen bits in front, then the items)
	const uint32_t hdr_dyn = readlane(wave_incl_scan(dyn), 63) + 3 + 5 + 5 + 4 + 3 * hclen;
	dyn = 0;
	for (uint32_t base = 0; base < 286; base += 64) {
		const uint32_t s = base + lane;
		if (s < 286) {
			const uint32_t f = lf[s];
			dyn += f * (lcode[s] >> 16);
			sta += f * (s < 144 ? 8u : s < 256 ? 9u : s < 280 ? 7u : 8u);
			if (s >= 265 && s < 285)
				extra += f * ((s - 261) >> 2);
		}
	}
	if (lane < 30) {
		const uint32_t f = df[lane];
		dyn += f * (dcode[lane] >> 16);
		sta += f * 5u;
		extra += f * (lane < 4 ? 0u : (lane >> 1) - 1);
	}
	dyn = readlane(wave_incl_scan(dyn), 63) + hdr_dyn;
	sta = readlane(wave_incl_scan(sta), 63) + 3;
	extra = readlane(wave_incl_scan(extra), 63);
	const bool use_dynamic = dyn < sta;                      // tie -> static (deflate_compress.c:1861-1867)
	EWB_T(4);
	if (!use_dynamic) {
		// the static code won (rare): its codes in place of the dynamic ones
		for (uint32_t s = lane; s < 288; s += 64) {
			const uint32_t len = s < 144 ? 8u : s < 256 ? 9u : s < 280 ? 7u : 8u;
			const uint32_t cw = s < 144 ? 0x30 + s : s < 256 ? 0x190 + (s - 144) : s < 280 ? s - 256 : 0xC0 + (s - 280);
			lcode[s] = (len << 16) | (__brev(cw) >> (32 - len));
		}
		if (lane < 32)
			dcode[lane] = (5u << 16) | (__brev(lane) >> 27);
	}
#pragma unroll
	for (uint32_t q = 0; q < 4; q++) {
		const uint32_t i = 64 * q + lane;                // literal i, length 3 + i
		const uint32_t lc = lcode[i];
		S->lut[i] = (lc & 0xffff) | (lc >> 16 << 24);
		uint32_t ls, leb, lev;
		len_slot(i + 3, ls, leb, lev);
		const uint32_t mc = lcode[257 + ls];
		S->lut[256 + i] = (mc & 0xffff) | (lev << (mc >> 16)) | (((mc >> 16) + leb) << 24);
	}
	if (lane == 0) {
		S->info[0] = (use_dynamic ? dyn : sta) + extra;
		S->info[1] = use_dynamic ? 1u : 0u;
		S->info[2] = use_dynamic ? hdr_dyn : 3u;
		S->info[3] = ni;
		S->info[4] = hlit;
		S->info[5] = hdist;
		S->info[6] = hclen;
	}
	EWB_T(5);
}

// ... and the bits of its header at `bstart`, its end-of-block code at the block's end (bbits on)
__device__ __forceinline__ void ew_emit_header(EW_LDS EwSlot *S, EW_LDS uint32_t *stage, uint32_t bstart, uint32_t bbits, bool final_bit, uint32_t lane)
{
	EW_LDS uint32_t *const pcode = S->L.pcode;
	EW_LDS uint16_t *const items = (EW_LDS uint16_t *)&S->build.hs.nf[64];
	uint32_t bitpos = bstart;
	if (uniform(S->info[1])) {
		const uint32_t ni = uniform(S->info[3]), hlit = uniform(S->info[4]), hdist = uniform(S->info[5]), hclen = uniform(S->info[6]);
		uint32_t c0 = 0, n0 = 0;
		if (lane == 0) { c0 = final_bit ? 1u : 0u; n0 = 1; }
		else if (lane == 1) { c0 = 2; n0 = 2; }
		else if (lane == 2) { c0 = hlit - 257; n0 = 5; }
		else if (lane == 3) { c0 = hdist - 1; n0 = 5; }
		else if (lane == 4) { c0 = hclen - 4; n0 = 4; }
		else if (lane < 5 + hclen) { c0 = pcode[k_perm19[lane - 5]] >> 16; n0 = 3; }
		bitpos += ew_emit1(stage, bitpos, c0, n0);
		for (uint32_t base = 0; base < ni; base += 64) {
			const uint32_t kx = base + lane;
			uint32_t cc = 0, nn = 0;
			if (kx < ni) {
				const uint32_t it = items[kx], sym = it & 31;
				const uint32_t pc = pcode[sym];
				cc = (pc & 0xffff) | ((it >> 8) << (pc >> 16));
				nn = (pc >> 16) + (sym == 16 ? 2u : sym == 17 ? 3u : sym == 18 ? 7u : 0u);
			}
			bitpos += ew_emit1(stage, bitpos, cc, nn);
		}
	} else {
		bitpos += ew_emit1(stage, bitpos, lane == 0 ? (final_bit ? 1u : 0u) : 1u, lane == 0 ? 1u : lane == 1 ? 2u : 0u);
	}
	const uint32_t eob = S->build.lcode[256];
	ew_emit1(stage, bstart + bbits - (eob >> 16), lane == 0 ? (eob & 0xffff) : 0u, lane == 0 ? (eob >> 16) : 0u);
}

#ifdef HD_EMIT_STATS
// experiment build only: wavefront 0's cycles by phase, g_emit_stats[0..7] (tools/exp_emit_wg_stats.py)
#define EW_T(k) do { if (threadIdx.x == 0) { const unsigned long long t_now = clock64(); atomicAdd(&g_emit_stats[k], t_now - t_ew); t_ew = t_now; } } while (0)
#else
#define EW_T(k) do { } while (0)
#endif

__global__ __launch_bounds__(64 * EW_NW) void k_emit_wg(DeflateArgs a)
{
	__shared__ EwLds E;
#ifdef HD_EMIT_STATS
	unsigned long long t_ew = clock64();
#endif
	const uint32_t lane = threadIdx.x & 63, w = uniform(threadIdx.x >> 6);
	const uint32_t bi = blockIdx.x, b = a.first + bi;
	const uint8_t *src = a.in + a.in_off[b];
	const uint32_t n = a.in_len[b];
	const SplitLayout lay = wg_layout(a.split_max);
	const uint8_t *rec = a.scratch + (uint64_t)bi * lay.bytes;
	const uint32_t *tok = (const uint32_t *)rec;
	const uint32_t *m = (const uint32_t *)(rec + lay.off_rec);
	const uint4 *pieces = (const uint4 *)(rec + lay.off_ntok);
	const uint32_t crcv = m[1];
	uint32_t *dst32 = (uint32_t *)(a.out + (uint64_t)b * a.out_stride);

	const uint32_t hdr = frame_hdr_bytes(a.frame), trl = frame_trl_bytes(a.frame), sfx = frame_sfx_bytes(a.frame);
	uint64_t cap64 = a.out_stride < a.out_cap ? a.out_stride : a.out_cap;
	if (a.frame == HD_FRAME_BGZF && cap64 > 65536)
		cap64 = 65536;
	const uint32_t cap = (uint32_t)cap64;
	const uint32_t stored = HD_STORED_SIZE(n);
	uint32_t limit = stored - 1;
	const bool flush = a.frame == HD_FRAME_RAW_FLUSH;
	bool alive = cap >= hdr + trl + sfx + 2;
	if (alive && cap - hdr - trl - sfx < limit)
		limit = cap - hdr - trl - sfx;
	const uint64_t limit_bits = 8ull * limit;
	if (m[0] != 0 || (a.wg_split > 1 && (m[2] & (0xffffffffu >> (32 - 8 * a.wg_split))) != 0))   // the parse refused the block or gave it up (any sharer): stored
		alive = false;
	const uint32_t np = alive ? (n + HD_WG_CUT - 1) / HD_WG_CUT : 0u;
	if (np > EW_MAX_PIECES) {                        // (launch_wg's promise: cannot happen; never run past the image)
		if (threadIdx.x == 0) {
			a.out_len[b] = 0;
			if (a.status) a.status[b] = 1;
			if (a.crc) a.crc[b] = crcv;
		}
		return;
	}

	// ---- the image zero, the length-symbol table --------------------------------------------------------------------
	for (uint32_t i = threadIdx.x; i < EW_STAGE_DW / 4; i += 64 * EW_NW)
		((uint4 *)E.stage)[i] = make_uint4(0, 0, 0, 0);
	if (w == 1) {
		uint32_t v = 0;
#pragma unroll
		for (uint32_t q = 0; q < 4; q++) {
			uint32_t ls, eb, ev;
			len_slot(4 * lane + q + 3, ls, eb, ev);
			v |= ls << (8 * q);
		}
		E.lsym[lane] = v;
	}
	__syncthreads();
	if (threadIdx.x < 4 && hdr)
		E.stage[threadIdx.x] = frame_hdr_word(a.frame, threadIdx.x);
	const uint8_t *const lsym = (const uint8_t *)E.lsym;

	// lane i: the record of piece i { tokens, literals, short matches, long matches }
	const uint4 pv = lane < np ? pieces[lane] : make_uint4(0, 0, 0, 0);
	const uint32_t paybase = 8 * hdr;
	uint32_t bitpos = paybase;

	// one field per lane (nbits <= 32, 0 = none), in lane order, into the image (wavefront 0: headers, suffix, trailer)
	auto put = [&](uint32_t code, uint32_t nbits, uint32_t bp) {
		if (nbits) {
			const uint32_t sh = bp & 31, i = bp >> 5;
			atomicOr(&E.stage[i], code << sh);
			if (sh + nbits > 32)
				atomicOr(&E.stage[i + 1], code >> (32 - sh));
		}
	};
	auto emit1 = [&](uint32_t code, uint32_t nbits) {
		const uint32_t incl = wave_incl_scan(nbits);
		put(code, nbits, bitpos + incl - nbits);
		bitpos += readlane(incl, 63);
	};
	// f(token of lane i, tokens in the group) over the tokens of pieces first, first + step, ... below end: the first 512
	// tokens of a piece are eight loads in flight, and the next piece's are requested before this one's are used (the
	// tokens lie in HBM / L2 where the parse left them: a round trip per 256 tokens was half of the passes' time)
	constexpr uint32_t PG = 8;
	auto for_tokens = [&](uint32_t first, uint32_t end, uint32_t step, auto &&enter, auto &&f, auto &&leave) {
		uint32_t cur[PG], nxt[PG];
		auto load = [&](uint32_t kk, uint32_t (&v)[PG]) {
			const uint32_t cnt = readlane(pv.x, kk);
			const uint32_t *pt = tok + kk * HD_WG_CUT;
#pragma unroll
			for (uint32_t j = 0; j < PG; j++)
				v[j] = 64 * j + lane < cnt ? pt[64 * j + lane] : 0u;
		};
		if (first < end)
			load(first, nxt);
		for (uint32_t kk = first; kk < end; kk += step) {
#pragma unroll
			for (uint32_t j = 0; j < PG; j++)
				cur[j] = nxt[j];
			if (kk + step < end)
				load(kk + step, nxt);
			const uint32_t cnt = readlane(pv.x, kk);
			enter(kk);
#pragma unroll
			for (uint32_t j = 0; j < PG; j++)
				if (64 * j < cnt)
					f(cur[j], cnt - 64 * j < 64 ? cnt - 64 * j : 64u);
			for (uint32_t base = 64 * PG; base < cnt; base += 64)       // a dense piece: the rest one load at a time
				f(base + lane < cnt ? tok[kk * HD_WG_CUT + base + lane] : 0u, cnt - base < 64 ? cnt - base : 64u);
			leave(kk);
		}
	};
	EW_T(0);

	// ---- wavefront 15 cuts the DEFLATE blocks (the emit-only kernel's scan, hd_deflate_dynamic.hpp; the twin's wg_split_check)
	// while the others count every piece's symbols into the piece's own histogram ---------------------------------------------
	if (w == EW_NW - 1) {
		// one v_readlane per piece: tokens | literals << 11 | short matches << 22 (each <= 1024)
		const uint32_t pk = pv.x | (pv.y << 11) | (pv.z << 22);
		uint32_t k = 0, block_begin = 0, nblk = 0;
		do {
			uint32_t obs0 = 0, obs1 = 0, obs2 = 0, sn = 0, no0 = 0, no1 = 0, no2 = 0, snn = 0, blk_tok = 0;
			bool end = false;
			while (k < np && !end) {
				const uint32_t q = readlane(pk, k);
				const uint32_t qt = q & 2047, ql = (q >> 11) & 2047, qs = q >> 22;
				blk_tok += qt;
				snn += qt;
				no0 += ql;
				no1 += qs;
				no2 += qt - ql - qs;
				k++;
				const uint32_t here = k * HD_WG_CUT;
				if (k < np) {
					end = blk_tok >= HD_DYN_BLOCK_TOKENS;
					if (!end && snn >= HD_WG_SPLIT_OBS && here - block_begin >= HD_WG_SPLIT_MIN && n - here >= HD_WG_SPLIT_MIN) {
						if (sn > 0) {
							const uint32_t e0 = obs0 * snn, a0 = no0 * sn, e1 = obs1 * snn, a1 = no1 * sn, e2 = obs2 * snn, a2 = no2 * sn;
							const uint32_t total = (a0 > e0 ? a0 - e0 : e0 - a0) + (a1 > e1 ? a1 - e1 : e1 - a1) + (a2 > e2 ? a2 - e2 : e2 - a2);
							const uint32_t items = sn + snn, blen = here - block_begin;
							uint32_t cutoff = snn * 200u / 512u * sn;
							if (blen < 10000 && items < 8192)
								cutoff += (cutoff >> 13) * (8192u - items);
							end = total + (blen / 4096u) * sn >= cutoff;
						}
						if (!end) {
							obs0 += no0;
							obs1 += no1;
							obs2 += no2;
							sn += snn;
							no0 = no1 = no2 = snn = 0;
						}
					}
				}
			}
			block_begin = k * HD_WG_CUT;
			if (lane == 0)
				E.cuts[1 + nblk] = k;
			nblk++;
		} while (k < np);
		if (lane == 0)
			E.cuts[0] = nblk;
	} else if (alive) {
		uint32_t *hp = nullptr;                                     // the histogram of the piece being counted
		for_tokens(w, np, EW_NW - 1,
			   [&](uint32_t kk) {
				   hp = E.ph[kk];
				   for (uint32_t i = lane; i < 160; i += 64)
					   hp[i] = 0;
			   },
			   [&](uint32_t tk, uint32_t nv) {
				   const bool is_match = (tk & HD_TOKEN_MATCH) != 0;
				   const uint32_t idx = (tk >> 16) & 0x1ffu;                // literal, or 256 + (length - 3)
				   const uint32_t sym = is_match ? 257u + lsym[idx & 0xffu] : idx;
				   uint32_t ds, eb, ev;
				   off_slot((tk & 0xffff) + 1, ds, eb, ev);
				   if (lane < nv) {
					   atomicAdd(&hp[sym >> 1], 1u << (16 * (sym & 1)));
					   if (is_match)
						   atomicAdd(&hp[144 + (ds >> 1)], 1u << (16 * (ds & 1)));
				   }
			   },
			   [&](uint32_t) {});
	}
	__syncthreads();
	EW_T(1);
	const uint32_t nblk = alive ? uniform(E.cuts[0]) : 0u;

	// ---- the DEFLATE blocks, EW_SLOTS at a time: their codes are built SIDE BY SIDE, a wavefront per block (wavefronts 0..3: one
	// per SIMD), because a member of 64 KiB is two or three blocks and a code construction is ~20 us of one wavefront's serial
	// work; a block's bits are known exactly from its histogram and code lengths, so every block, header and piece has its place
	// in the image before a single token is coded ------------------------------------------------------------------------------
	for (uint32_t r0 = 0; r0 < nblk && alive; r0 += EW_SLOTS) {
		const uint32_t nb = nblk - r0 < EW_SLOTS ? nblk - r0 : EW_SLOTS;               // blocks of this round
		const uint32_t kr0 = r0 ? uniform(E.cuts[r0]) : 0u;                             // the round's first piece
		uint32_t kend[EW_SLOTS];                                                        // the piece behind block i of the round
#pragma unroll
		for (uint32_t i = 0; i < EW_SLOTS; i++)
			kend[i] = uniform(E.cuts[1 + r0 + (i < nb ? i : nb - 1)]);
		// ---- the blocks' symbols: the sums of their pieces' (packed adds: a block holds < 2^16 tokens) -------------------------
		{
			const uint32_t i = threadIdx.x / 160, t = threadIdx.x % 160;
			if (i < nb) {
				const uint32_t k0 = i == 0 ? kr0 : i == 1 ? kend[0] : i == 2 ? kend[1] : kend[2];
				const uint32_t k1 = i == 0 ? kend[0] : i == 1 ? kend[1] : i == 2 ? kend[2] : kend[3];
				uint32_t acc = 0;
				for (uint32_t kk = k0; kk < k1; kk++)
					acc += E.ph[kk][t];
				EW_LDS DynLds *Ls = &((EW_LDS EwSlot *)&E.slot[0])[i].L;
				if (t < 144) {
					Ls->lf[2 * t] = acc & 0xffffu;
					Ls->lf[2 * t + 1] = acc >> 16;
				} else {
					Ls->df[2 * (t - 144)] = acc & 0xffffu;
					Ls->df[2 * (t - 144) + 1] = acc >> 16;
				}
			}
		}
		__syncthreads();
		EW_T(2);
		// ---- wavefront i < nb: the codes of block r0 + i, the RLE of their lengths, the precode, the exact cost, the choice
		// (flush_block of the emit-only kernel, statement by statement), the litlen table -- everything but the bits -------------
		// (its own function, the slot through an LDS-typed pointer: with `E.slot[w]` inline the compiler loses track of the
		// address space and reaches LDS through flat instructions -- the serial RLE / cost code ran four times slower so --,
		// and four inlined copies spill)
		// The litlen code on wavefront i, the offset code on wavefront 4 + i (another SIMD's: tools/simd_probe.hip), its scratch
		// lent from the image, which is all zeros and nobody's until the headers go in -- and is zeroed again below
		static_assert(EW_SCRATCH_DW * 4 >= sizeof(HuffScratch) && 1024 + EW_SLOTS * EW_SCRATCH_DW <= EW_STAGE_DW, "the lent scratch lies inside the image");
		// (only in the first round: behind it the image holds the blocks already coded, and a member of more than four
		// DEFLATE blocks -- rare -- builds the later ones' two codes one after the other)
		const bool lend = r0 == 0;
		if (w < nb) {
			ew_build_code((EW_LDS EwSlot *)&E.slot[0] + w, 0, &((EW_LDS EwSlot *)&E.slot[0] + w)->build.hs, lane);
			if (!lend)
				ew_build_code((EW_LDS EwSlot *)&E.slot[0] + w, 1, &((EW_LDS EwSlot *)&E.slot[0] + w)->build.hs, lane);
		} else if (lend && w >= 4 && w - 4 < nb) {
			ew_build_code((EW_LDS EwSlot *)&E.slot[0] + (w - 4), 1, (EW_LDS HuffScratch *)((EW_LDS uint32_t *)E.stage + 1024 + (w - 4) * EW_SCRATCH_DW), lane);
		}
		__syncthreads();
		if (lend)
			for (uint32_t i = threadIdx.x; i < EW_SLOTS * EW_SCRATCH_DW; i += 64 * EW_NW)
				E.stage[1024 + i] = 0;
		if (w < nb)
			ew_prepare_block((EW_LDS EwSlot *)&E.slot[0] + w, lane);
		__syncthreads();
		EW_T(3);
		// ---- every block of the round has its place (and the member its verdict: a block that does not fit ends it) ------------
		uint32_t bstart[EW_SLOTS], bbits[EW_SLOTS];
		{
			uint32_t at = bitpos;
#pragma unroll
			for (uint32_t i = 0; i < EW_SLOTS; i++) {
				bstart[i] = at;
				bbits[i] = i < nb ? uniform(E.slot[i].info[0]) : 0u;
				if (i < nb && (uint64_t)(at - paybase) + bbits[i] > limit_bits)
					alive = false;
				at += bbits[i];
			}
			bitpos = at;
		}
		if (!alive)
			break;
		// ---- wavefront i < nb: the header of its block; the others: what every piece of the round weighs, from its histogram and
		// the code lengths of its block ------------------------------------------------------------------------------------------
		const EW_LDS EwSlot *const slots3 = (const EW_LDS EwSlot *)&E.slot[0];     // (LDS pointers stay LDS pointers: ds_read, not flat_load)
		if (w < nb) {
			const uint32_t bs = w == 0 ? bstart[0] : w == 1 ? bstart[1] : w == 2 ? bstart[2] : bstart[3];
			const uint32_t bb = w == 0 ? bbits[0] : w == 1 ? bbits[1] : w == 2 ? bbits[2] : bbits[3];
			ew_emit_header((EW_LDS EwSlot *)&E.slot[0] + w, (EW_LDS uint32_t *)E.stage, bs, bb, r0 + w + 1 == nblk && !flush, lane);
		} else {
			for (uint32_t kk = kr0 + (w - nb); kk < kend[EW_SLOTS - 1]; kk += EW_NW - nb) {
				const uint32_t si = (kk >= kend[0] ? 1u : 0u) + (kk >= kend[1] ? 1u : 0u) + (kk >= kend[2] ? 1u : 0u);
				const EW_LDS uint32_t *lc = slots3[si].build.lcode, *dc = slots3[si].build.dcode;
				uint32_t bits = 0;
				for (uint32_t i = lane; i < 160; i += 64) {
					const uint32_t c = E.ph[kk][i];
					uint32_t n0, n1;                                   // bits of one symbol 2 i / 2 i + 1: codeword + extra bits
					if (i < 144) {
						const uint32_t s0 = 2 * i, s1 = 2 * i + 1;
						n0 = (lc[s0] >> 16) + ((s0 >= 265 && s0 < 285) ? (s0 - 261) >> 2 : 0u);
						n1 = (lc[s1] >> 16) + ((s1 >= 265 && s1 < 285) ? (s1 - 261) >> 2 : 0u);
					} else {
						const uint32_t d0 = 2 * (i - 144), d1 = d0 + 1;
						n0 = (dc[d0] >> 16) + (d0 < 4 ? 0u : (d0 >> 1) - 1);
						n1 = (dc[d1] >> 16) + (d1 < 4 ? 0u : (d1 >> 1) - 1);
					}
					bits += (c & 0xffffu) * n0 + (c >> 16) * n1;
				}
				bits = readlane(wave_incl_scan(bits), 63);
				if (lane == 0)
					E.piece_bits[kk] = bits;
			}
		}
		__syncthreads();
		EW_T(4);
		// ---- every piece has its place; the pieces of the round are coded side by side ----------------------------------------
		const uint32_t pb = (lane >= kr0 && lane < kend[EW_SLOTS - 1]) ? E.piece_bits[lane] : 0u;
		const uint32_t pex = wave_incl_scan(pb) - pb;                                   // bits of the round's pieces in front of piece `lane`
		uint32_t bp = 0;
		const EW_LDS uint32_t *lut = nullptr, *dcode = nullptr;
		for_tokens(kr0 + w, kend[EW_SLOTS - 1], EW_NW,
			   [&](uint32_t kk) {
				   const uint32_t si = (kk >= kend[0] ? 1u : 0u) + (kk >= kend[1] ? 1u : 0u) + (kk >= kend[2] ? 1u : 0u);
				   const uint32_t kb = si == 0 ? kr0 : si == 1 ? kend[0] : si == 2 ? kend[1] : kend[2];    // the block's first piece
				   const uint32_t bs = si == 0 ? bstart[0] : si == 1 ? bstart[1] : si == 2 ? bstart[2] : bstart[3];
				   bp = bs + uniform(slots3[si].info[2]) + readlane(pex, kk) - readlane(pex, kb);
				   lut = slots3[si].lut;
				   dcode = slots3[si].build.dcode;
			   },
			   [&](uint32_t tk, uint32_t nv) {
				   const bool valid = lane < nv;
				   const bool is_match = (tk & HD_TOKEN_MATCH) != 0;
				   const uint32_t le = lut[(tk >> 16) & 0x1ffu];
				   const uint32_t ca = le & 0xffffffu, na = le >> 24;
				   uint32_t ds, deb, dev;
				   off_slot((tk & 0xffff) + 1, ds, deb, dev);
				   const uint32_t dc = dcode[ds];
				   const uint32_t cb = (dc & 0xffff) | (dev << (dc >> 16));
				   const uint32_t nb2 = is_match ? (dc >> 16) + deb : 0u;
				   const uint32_t nbits = valid ? na + nb2 : 0u;
				   const uint64_t code = valid ? ((uint64_t)ca | ((uint64_t)(is_match ? cb : 0u) << na)) : 0ull;
				   const uint32_t incl = wave_incl_scan(nbits);
				   const uint32_t at = bp + incl - nbits;
				   const uint32_t sh = at & 31, i = at >> 5;
				   const uint64_t lo = code << sh;                      // bits [0, 64) of the shifted field
				   const uint32_t hi = sh ? (uint32_t)(code >> (64 - sh)) : 0u;   // and what a 48-bit field spills beyond
				   atomicOr(&E.stage[i], (uint32_t)lo);
				   atomicOr(&E.stage[i + 1], (uint32_t)(lo >> 32));
				   atomicOr(&E.stage[i + 2], hi);
				   bp += readlane(incl, 63);
			   },
			   [&](uint32_t) {});
		EW_T(5);
		if (r0 + EW_SLOTS < nblk)
			__syncthreads();                                    // (the next round's sums and codes take the slots' places)
	}

	__syncthreads();
	EW_T(6);
	if (!alive) {
		if (w == 0)
			write_stored_member(a, b, src, n, crcv, lane);
		return;
	}
	if (w == 0) {
		if (flush) {
			// empty stored block header (000), alignment, LEN = 0, NLEN = ffff
			bitpos = (bitpos + 3 + 7) & ~7u;
			emit1(lane == 1 ? 0xffffu : 0u, lane < 2 ? 16u : 0u);
		}
		bitpos = (bitpos + 7) & ~7u;
		const uint32_t paylen = (bitpos - paybase) >> 3;
		if (trl) {
			uint32_t tcode = 0, nb = 0;
			if (lane < trl / 2) {
				tcode = frame_trl_field(a.frame, lane, crcv, n);
				nb = 16;
			}
			emit1(tcode, nb);
		}
		if (lane == 0) {
			const uint32_t total = hdr + paylen + trl;
			if (a.frame == HD_FRAME_BGZF)
				atomicOr(&E.stage[4], (total - 1) & 0xffffu);          // BSIZE: bytes 16..17 of the header
			else if (a.frame == HD_FRAME_MIGZ)
				E.stage[4] = paylen;
			E.ctl[3] = total;
			a.out_len[b] = total;
			if (a.status) a.status[b] = 0;
			if (a.crc) a.crc[b] = crcv;
		}
	}
	__syncthreads();
	const uint32_t total = uniform(E.ctl[3]);
	const uint32_t q16 = total >> 4;
	for (uint32_t i = threadIdx.x; i < q16; i += 64 * EW_NW)
		((uint4 *)dst32)[i] = ((const uint4 *)E.stage)[i];
	if (threadIdx.x < ((total + 3) >> 2) - 4 * q16)
		dst32[4 * q16 + threadIdx.x] = E.stage[4 * q16 + threadIdx.x];
	EW_T(7);
}

inline void launch_emit_wg(const DeflateArgs &s, hipStream_t st)
{
	hipLaunchKernelGGL(k_emit_wg, dim3(s.count), dim3(64 * EW_NW), 0, st, s);
}

} // namespace hd
