// hd_inflate_lat.hpp -- the LATENCY form of the decoder (hip_inflate / hip_inflate_flush: one stream per call, the callers
// waiting -- lib/zlibutil.c:194-204 as applet/7bgzf.c:330-345 calls it, a thread per block): FOUR wavefronts per stream.
//
// The same decoder as hd_inflate.hpp's inflate_stream -- same tables, same windows, same scalar path, the same verdicts -- with
// its statements dealt to four wavefronts of one workgroup, each on a SIMD of its own (the comments at inflate_stream_pipe):
//   spec   decodes the token that WOULD start at every bit position of the stream, ahead of everybody (an LDS ring of words)
//   front  owns the bit reader, the headers and tables, every verdict: reads the words under its window, walks the real chain,
//          sums the output positions -- what decides where the next window starts -- and hands the rest over
//   sort   stores the window's literals, sorts its matches by the way they are copied, writes the record
//   back   copies the matches, flushes the ring, folds the CRC, writes the verdict
// It is a copy of that function, restructured, and not a template parameter of it: the throughput kernel is bound by vector
// issue at six wavefronts per SIMD, three registers below the step that costs one, and the same restructure applied to it in
// place -- placement behind a lambda, then inline again with only the scalar writers and the tail as lambdas -- cost it 6 % and
// 4 % on one box (profiles/r05_inflate_ab.txt).  And this copy is free to be what the latency kernel wants: its ring holds the
// whole window of DEFLATE, so the "far" paths (sources that left the ring) are gone.  A lone call of a 0xff00-byte block:
// 1.56 ms as one wavefront (round 4), 1.12 as front + back, 1.00 with spec, 0.86-0.90 with sort.
#pragma once
#include "hd_inflate.hpp"

namespace hd {

// PIPE (k_inflate_lat): the dump slots are their own bytes -- the front wavefront parses a block
// header in cl / pre_lens while the back one still places the window before it -- and the records the front hands the back
constexpr uint32_t INF_PQ = 4;               // window records in flight between the sort and the back wavefront
constexpr uint32_t INF_SPEC_POS = 2048;      // bit positions whose speculative decode is kept: 32 chunks of 64
constexpr uint32_t INF_SPEC_CHUNKS = INF_SPEC_POS / 64;
constexpr uint32_t INF_LAT_THREADS = 256;    // front, back, spec, sort
constexpr uint32_t INF_LAT_SPINS = 1u << 24;  // polls of a wait between the wavefronts before the stream is given up (~0.2 s; a wait is microseconds):
                                             // never a hang -- hip_inflate answers HD_BAD_DATA, and no test has seen it happen.  (The guards and
                                             // the skipped distance check measured neutral on one box: 898-904 us a lone call either way)
constexpr uint32_t INF_FQ = 4;               // windows in flight between the front and the sort wavefront
template <uint32_t RING>
struct InfLdsPipeT {
	uint32_t lit[1u << INF_LT_BITS];
	uint32_t off[1u << INF_DT_BITS];
	uint16_t lit_sorted[288];
	uint16_t off_sorted[32];
	uint16_t lit_count[16], off_count[16];
	union {
		__attribute__((aligned(16))) uint8_t ring[RING];
		uint32_t ring32[RING / 4];
	};
	uint8_t dump[64];                        // right behind the ring: ring[RING + lane]
	uint8_t cl[288 + 32 + 138 + 6];
	uint8_t pre_lens[32];
	uint32_t comp[128];
	// front -> back: head / tail count records; hdr: { type, pos | x, cum | y, four lane masks }; per lane three dwords
	uint32_t q_head, q_tail;
	uint32_t q_hdr[INF_PQ][12];             // (eleven words used; the struct stays within half a CU's LDS: tests/test_abi.py)
	uint32_t q_lane[INF_PQ][3][64];
	// spec -> front: one word per BIT POSITION of the stream (position mod INF_SPEC_POS), chunks of 64 positions; the words
	// that count chunks and pass the tables between the two (inflate_stream_pipe)
	uint32_t spec[INF_SPEC_POS];
	uint32_t sp_head, sp_tail, sp_stop, sp_ack, sp_go, sp_start;
	// front -> sort: { type, x, y, B, the two masks of the tokens the walk took }
	uint32_t fq_head, fq_tail;
	uint32_t fq_hdr[INF_FQ][8];
	uint32_t abort;                          // a wait between the wavefronts did not end (INF_LAT_SPINS): everybody leaves, the verdict is an error
};

// the decoder, for an output ring of RING bytes (one wavefront; L is the workgroup's LDS).
// PIPE (the latency kernel), first step: TWO wavefronts per stream (spec and sort came later, below).  One wavefront's decode of a stream is a chain of latencies -- table
// gathers, the scalar walk, LDS permutes, ring round trips: 1.55 ms for a 0xff00-byte block however empty the chip -- of which
// the part that FINDS the tokens (speculative decode, walk, prefix sum, the checks) needs nothing of the part that PLACES them
// (literals, lane-group copies, match copies, flush, CRC).  So the front wavefront (threads 0..63) owns the bit reader, the
// tables and every verdict, and keeps the output position as a number only; whatever writes the ring -- a window's tokens, a
// literal or match of the scalar path, a stored block -- goes to the back wavefront (threads 64..127) as a record through
// LDS, INF_PQ in flight.  Statement for statement the decoder of hd_inflate.hpp: only who executes which half differs.
enum { PIPE_WINDOW = 0, PIPE_LITERAL = 1, PIPE_MATCH = 2, PIPE_STORED = 3, PIPE_END = 4 };
template <uint32_t RING, class LDS>
__device__ __forceinline__ void inflate_stream_pipe(const InflateArgs &a, LDS &L)
{
	constexpr uint32_t INF_NEAR = RING - 258 - 64;       // dist <= this: source is in the ring
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t role = threadIdx.x >> 6;              // 0 front, 1 back, 2 spec, 3 sort
	const bool back = role == 1;
	if (threadIdx.x == 0) {
		L.q_head = L.q_tail = 0;
		L.fq_head = L.fq_tail = 0;
		L.abort = 0;
		L.sp_head = L.sp_tail = L.sp_ack = L.sp_go = L.sp_start = 0;
		L.sp_stop = 1;                                   // the spec wavefront starts halted: there are no tables yet (epoch 1 = the first header's)
	}
	__syncthreads();
	// every wait of one wavefront for another goes through this: `pending()` true = keep waiting
	typedef volatile __attribute__((address_space(3))) uint32_t *lat_word_p;
	auto lat_wait = [&](auto &&pending, uint32_t nap) {
		for (uint32_t spins = 0; pending(); spins++) {
			// (looked at every 256th poll: the polls themselves must stay as short as they were)
			if ((spins & 255u) == 255u && (spins > INF_LAT_SPINS || uniform(*(lat_word_p)&L.abort))) {
				if (lane == 0)
					*(lat_word_p)&L.abort = 1;
				break;
			}
			if (nap)
				__builtin_amdgcn_s_sleep(1);
			else
				__builtin_amdgcn_s_sleep(0);
		}
	};
#ifdef HD_INFLATE_STATS
	// experiment build (tools/exp_inflate_pipe_stats.py): a wavefront's cycles in all, and those it waited for the other one
	unsigned long long pipe_wait = 0;
	const unsigned long long pipe_t0 = __builtin_amdgcn_s_memtime();
#define PIPE_W0(t) const unsigned long long t = __builtin_amdgcn_s_memtime()
#define PIPE_W1(t) do { pipe_wait += __builtin_amdgcn_s_memtime() - (t); } while (0)
	unsigned long long pipe_acc[6] = { 0, 0, 0, 0, 0, 0 };     // the front's window loop by phase
	unsigned long long pipe_tp = 0;
#define PIPE_P0() do { pipe_tp = __builtin_amdgcn_s_memtime(); } while (0)
#define PIPE_P(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pipe_acc[k] += t_ - pipe_tp; pipe_tp = t_; } while (0)
#define PIPE_WFLUSH() do { if (!back) for (int k_ = 0; k_ < 6; k_++) atomicAdd(&g_inf_cycles[k_], lane == 0 ? pipe_acc[k_] : 0ull); \
		atomicAdd(&g_inf_stats2[back ? 7 : 6], lane == 0 ? pipe_wait : 0ull); \
		atomicAdd(&g_inf_stats[back ? 7 : 6], lane == 0 ? __builtin_amdgcn_s_memtime() - pipe_t0 : 0ull); } while (0)
#else
#define PIPE_W0(t) do { } while (0)
#define PIPE_W1(t) do { } while (0)
#define PIPE_P0() do { } while (0)
#define PIPE_P(k) do { } while (0)
#define PIPE_WFLUSH() do { } while (0)
#endif
	const uint32_t b = blockIdx.x;
	if (b >= a.nblocks)
		return;
	const ClockStamp clk(HD_CLK_INFLATE);
	const uint8_t *src = a.in + a.in_off[b];
	const uint32_t n = a.in_len[b];
	if (n >= HD_INFLATE_MAX_IN) {
		// stream positions are 32-bit BIT counts (over_t, B below): a stream this long is refused whole
		// rather than decoded from wrapped positions (the host entry points answer HD_E_ARG before launching)
		if (threadIdx.x == 0) {
			a.out_len[b] = 0;
			if (a.crc) a.crc[b] = 0;
			if (a.status) a.status[b] = HD_BAD_DATA;
		}
		return;
	}
	uint8_t *dst = a.out + a.out_off[b];
	const uint32_t cap = a.out_cap[b];
	const CrcTables *ct = a.ct;
	const bool want_crc = a.crc != nullptr;
	const bool dst_aligned = (((uintptr_t)dst) & 15) == 0;

	// ---- compressed input: 256-byte pieces, one dword per lane -----------
	const uint32_t mis = (uint32_t)((uintptr_t)src & 3);
	const uint32_t *src32 = (const uint32_t *)(src - mis);
	const uint32_t nbytes_al = mis + n;                  // valid bytes from src32
	auto load_piece = [&](uint32_t piece) -> uint32_t {
		const uint32_t d = piece * 64 + lane;
		uint32_t w = 0;
		if (d * 4 < nbytes_al) {
			w = src32[d];
			const uint32_t valid = nbytes_al - d * 4;    // bytes of this dword inside the stream
			if (valid < 4)
				w &= (1u << (8 * valid)) - 1;
		}
		return w;
	};
	uint32_t cur_piece = 0;
	uint32_t cw = load_piece(0), cw_next = load_piece(1);
	uint32_t dw = 0;                 // next dword index to feed the bit buffer
	uint64_t bb = 0;                 // bit buffer (uniform)
	uint32_t bc = 0;                 // valid bits in bb

	auto next_dword = [&]() -> uint32_t {
		const uint32_t piece = dw >> 6;
		if (piece != cur_piece) {        // uniform branch: step to the next piece
			cw = cw_next;
			cur_piece = piece;
			cw_next = load_piece(piece + 1);
		}
		const uint32_t w = readlane(cw, dw & 63);
		dw++;
		return w;
	};
	auto refill = [&]() {
		if (bc <= 32) {
			bb |= (uint64_t)next_dword() << bc;
			bc += 32;
		}
	};
	auto consumed_bits = [&]() -> int64_t { return (int64_t)dw * 32 - bc - 8 * (int64_t)mis; };
	// consumed_bits() > 8 n + 64 in 32-bit arithmetic (n < 2^28): (dw << 5) - bc > over_t
	const uint32_t over_t = 8 * n + 64 + 8 * mis;
	auto overrun = [&]() -> bool { return (dw << 5) - bc > over_t; };
	auto seek_byte = [&](uint32_t byteoff) {       // restart the bit reader at src + byteoff
		const uint32_t o = mis + byteoff;
		dw = o >> 2;
		const uint32_t piece = dw >> 6;
		if (piece != cur_piece) {
			cur_piece = piece;
			cw = load_piece(piece);
			cw_next = load_piece(piece + 1);
		}
		bb = 0;
		bc = 0;
		refill();
		bb >>= 8 * (o & 3);
		bc -= 8 * (o & 3);
	};
	// skip the mis-alignment bytes
	refill();
	bb >>= 8 * mis;
	bc -= 8 * mis;

	// ---- output ring + flush ---------------------------------------------
	uint32_t pos = 0, flushed = 0;
	CrcLanes crc;
	crc.init(lane, 0xffffffffu);          // length unknown yet; lane 0 seeds, fixed in finish
	auto flush_pieces = [&]() {
		while (pos - flushed >= HD_PIECE) {
			const uint4 v = *(const uint4 *)&L.ring[(flushed & (RING - 1)) + 16 * lane];
			if (dst_aligned) {
				*(uint4 *)(dst + flushed + 16 * lane) = v;
			} else {
				const uint32_t w[4] = { v.x, v.y, v.z, v.w };
				for (uint32_t k = 0; k < 16; k++)
					dst[flushed + 16 * lane + k] = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
			}
			if (want_crc)
				crc.fold(ct, flushed / HD_PIECE, true, v);
			flushed += HD_PIECE;
		}
	};


	// ---- window decode: 64 speculative tokens per pass -----------------------
	// The scalar token loop costs ~80 SALU per token on the CU's single scalar ALU.
	// Here every lane decodes the token that WOULD start at bit B + lane (two LDS
	// table gathers, VALU only); a short scalar walk then follows the real chain
	// (~10 SALU per token), literals are stored in parallel at positions from a
	// DPP prefix sum, matches are copied in order.  A window is entered only
	// when nothing rare can happen inside it: five whole dwords of stream ahead,
	// some room in the output (the window's budget), less than a piece waiting for the flush.
	// Returns 0 = fall back to the scalar loop for one token, 1 = end of block
	// consumed, 2 = error (st set).  Reader state is the scalar one on both sides.
	// output budget of one window: pending <= 1023 + 704 = 1727 <= RING - 64 - 257, so a source is
	// either wholly in the ring (wend - src <= RING - 64) or wholly flushed to HBM
	constexpr uint32_t WIN_OUT_BUDGET = 704;
	const uint32_t dw_safe = (mis + n) >> 2;      // dwords below this are whole

	// ---- the scalar path's writers (PIPE: the back wavefront's) -----------------------------------------------------------------
	auto copy_stored = [&](uint32_t ip, uint32_t len) {
		for (uint32_t done = 0; done < len;) {
			const uint32_t step = len - done < 64 ? len - done : 64;
			if (lane < step)
				L.ring[(pos + lane) & (RING - 1)] = src[ip + done + lane];
			pos += step;
			done += step;
			if (pos - flushed >= HD_PIECE)
				flush_pieces();
		}
	};
	auto copy_match = [&](uint32_t length, uint32_t offset) {
		// ---- match copy, 64 bytes per step ------------------------
		if (offset <= INF_NEAR) {
			if (offset >= length) {
				// disjoint: the common case, one pass per 64 bytes
				for (uint32_t i = lane; i < length; i += 64)
					L.ring[(pos + i) & (RING - 1)] = L.ring[(pos - offset + i) & (RING - 1)];
			} else {
				// overlapping (run of period `offset`): source index i mod offset,
				// by a uniform reciprocal; all sources lie before pos
				const float rcp = 1.0f / (float)offset;
				for (uint32_t i = lane; i < length; i += 64) {
					uint32_t q = (uint32_t)((float)i * rcp);
					uint32_t r = i - q * offset;
					r = (int32_t)r < 0 ? r + offset : r;
					r = r >= offset ? r - offset : r;
					L.ring[(pos + i) & (RING - 1)] = L.ring[(pos - offset + r) & (RING - 1)];
				}
			}
		} else {
			// source was flushed long ago: make our own stores visible
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll 1
			for (uint32_t i = lane; i < length; i += 64)
				L.ring[(pos + i) & (RING - 1)] = dst[pos - offset + i];
		}
	};

	// ---- what a window's MATCHES become in the ring: the back wavefront's half of a window.  The front has stored the literals
	// itself (they depend on nothing) and has sorted the matches -- none of that needs the ring --; here the ring holds every
	// source (64 KiB against distances <= 32 KiB + the few KiB the front is ahead: the "far" paths of hd_inflate.hpp's decoder
	// do not exist), so what is left is the copying: the statements of that decoder's window, in its order -------------------
	struct WinRec {
		uint32_t lo0, lo1;                   // per lane and half: match length | distance << 16
		uint32_t rel0, rel1;                 // a real token's first output byte, relative to the window's
		// lane masks of the matches (the back wavefront sorts the simple ones by length: <= 8 bytes lane groups of eight, 9..16
		// groups of sixteen, 17..64 one at a time):
		uint64_t simple0, simple1;           // source wholly in front of the window's output, <= 64 bytes: nothing in the window feeds them
		uint64_t g0, g1;                     // the others -- fed by the window's own output, overlapping, longer -- in stream order
		uint32_t pos, cum;                   // the window's first output byte, its output bytes
	};
	auto place_window = [&](const WinRec &W) {
		const uint32_t pos = W.pos;          // (shadows the decoder's: this window's)
		const uint32_t wend = pos + W.cum;
		const uint32_t len0 = W.lo0 & 0xffffu, off0 = W.lo0 >> 16, len1 = W.lo1 & 0xffffu, off1 = W.lo1 >> 16;
		const uint32_t opos0 = pos + W.rel0, opos1 = pos + W.rel1;
		const uint32_t srcl0 = opos0 - off0, srcl1 = opos1 - off1;
		const uint64_t l8_0 = __ballot(len0 <= 8), l8_1 = __ballot(len1 <= 8);
		const uint64_t l16_0 = __ballot(len0 <= 16), l16_1 = __ballot(len1 <= 16);
		const uint64_t a0 = W.simple0 & l8_0, a1 = W.simple1 & l8_1;
		const uint64_t b0 = W.simple0 & l16_0 & ~l8_0, b1 = W.simple1 & l16_1 & ~l8_1;
		bool pend = false;                   // a lane-group pass is open: bytes pend_v for ring[pend_idx]
		uint32_t pend_idx = 0, pend_v = 0;
		if (a0 | a1 | b0 | b1) {
			// LANE GROUPS (hd_inflate.hpp): eight (sixteen) lanes per match, eight (four) matches per pass, no scalar work per
			// match -- every owner pushes {offset in the window's output, length, distance} to the first lane of its group
			// (ds_permute), the group fetches it (ds_bpermute), each lane moves one byte.  rel < 1024, length <= 16, distance
			// <= 32768: 10 + 5 + 16 bits.  The first pass of a window stays open across the scalar copies below.
			const uint32_t pk0 = W.rel0 | (len0 << 10) | (off0 << 15), pk1 = W.rel1 | (len1 << 10) | (off1 << 15);
			auto group_pass = [&](auto gtag, uint64_t own0, uint64_t own1) {
				constexpr uint32_t G = decltype(gtag)::value, NG = 64 / G;
				const uint32_t n0 = (uint32_t)__popcll(own0), nt = n0 + (uint32_t)__popcll(own1);
				const uint32_t slot0 = __builtin_amdgcn_mbcnt_hi((uint32_t)(own0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)own0, 0));
				const uint32_t slot1 = n0 + __builtin_amdgcn_mbcnt_hi((uint32_t)(own1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)own1, 0));
				const uint32_t sub = lane & (G - 1), lead = (lane & ~(G - 1)) << 2;
				for (uint32_t base = 0; base < nt; base += NG) {
					// an owner whose slot falls into this pass targets the first lane of group (slot - base); everybody
					// else an odd lane (never a group's first): what arrives there is not looked at
					const uint32_t d0 = sel(own0, slot0 - base, NG), d1 = sel(own1, slot1 - base, NG);
					const uint32_t t0 = d0 < NG ? d0 * (4 * G) : ((lane | 1u) << 2), t1 = d1 < NG ? d1 * (4 * G) : ((lane | 1u) << 2);
					const uint32_t g0 = (uint32_t)__builtin_amdgcn_ds_permute((int)t0, (int)(d0 < NG ? pk0 : 0u));
					const uint32_t g1 = (uint32_t)__builtin_amdgcn_ds_permute((int)t1, (int)(d1 < NG ? pk1 : 0u));
					const uint32_t w = (uint32_t)__builtin_amdgcn_ds_bpermute((int)lead, (int)(g0 | g1));
					const uint32_t ml = (w >> 10) & 31;                        // 0: no match in this group
					const uint32_t dp = pos + (w & 1023) + sub, sp = dp - (w >> 15);
					const uint32_t v = L.ring[sp & (RING - 1)];
					const uint32_t di = sub < ml ? (dp & (RING - 1)) : RING + lane;
					if (!pend) {
						pend = true;
						pend_idx = di;
						pend_v = v;
					} else {
						L.ring[di] = (uint8_t)v;
					}
				}
			};
			if (a0 | a1)
				group_pass(std::integral_constant<uint32_t, 8>{}, a0, a1);
			if (b0 | b1)
				group_pass(std::integral_constant<uint32_t, 16>{}, b0, b1);
		}
		// the other simple ones, one at a time
		for (uint64_t sm = W.simple0 & ~l16_0; sm;) {
			const uint32_t m = (uint32_t)__ffsll((unsigned long long)sm) - 1;
			asm("s_bitset0_b64 %0, %1" : "+s"(sm) : "s"(m));       // (sm &= sm - 1 is three scalar instructions)
			const uint32_t mlen = readlane(len0, m), P = readlane(opos0, m), srcp = readlane(srcl0, m);
			const uint8_t v = L.ring[(srcp + lane) & (RING - 1)];
			L.ring[lane < mlen ? ((P + lane) & (RING - 1)) : RING + lane] = v;
		}
		for (uint64_t sm = W.simple1 & ~l16_1; sm;) {
			const uint32_t m = (uint32_t)__ffsll((unsigned long long)sm) - 1;
			asm("s_bitset0_b64 %0, %1" : "+s"(sm) : "s"(m));
			const uint32_t mlen = readlane(len1, m), P = readlane(opos1, m), srcp = readlane(srcl1, m);
			const uint8_t v = L.ring[(srcp + lane) & (RING - 1)];
			L.ring[lane < mlen ? ((P + lane) & (RING - 1)) : RING + lane] = v;
		}
		if (pend)
			L.ring[pend_idx] = (uint8_t)pend_v;
		// the general ones, in stream order (the literals and the simple matches of the whole window are in)
		auto copy_general = [&](uint32_t mlen, uint32_t P, uint32_t srcp) {
			const uint32_t moff = P - srcp;
			if (moff >= mlen) {
				for (uint32_t i = lane; i < mlen; i += 64)
					L.ring[(P + i) & (RING - 1)] = L.ring[(srcp + i) & (RING - 1)];
			} else {
				const float rcp = 1.0f / (float)moff;
				for (uint32_t i = lane; i < mlen; i += 64) {
					uint32_t q = (uint32_t)((float)i * rcp);
					uint32_t r = i - q * moff;
					r = (int32_t)r < 0 ? r + moff : r;
					r = r >= moff ? r - moff : r;
					L.ring[(P + i) & (RING - 1)] = L.ring[(srcp + r) & (RING - 1)];
				}
			}
		};
		for (uint64_t mm = W.g0; mm;) {
			const uint32_t m = (uint32_t)__ffsll((unsigned long long)mm) - 1;
			asm("s_bitset0_b64 %0, %1" : "+s"(mm) : "s"(m));
			copy_general(readlane(len0, m), readlane(opos0, m), readlane(srcl0, m));
		}
		for (uint64_t mm = W.g1; mm;) {
			const uint32_t m = (uint32_t)__ffsll((unsigned long long)mm) - 1;
			asm("s_bitset0_b64 %0, %1" : "+s"(mm) : "s"(m));
			copy_general(readlane(len1, m), readlane(opos1, m), readlane(srcl1, m));
		}
		(void)wend;
	};

	// ---- PIPE: the records to the back wavefront.  The LDS executes a wavefront's instructions in order, so the store of
	// q_head behind a record's stores publishes the record (the protocol of k_parse_wg's turns, hd_deflate_wg.hpp) ----------
	typedef volatile __attribute__((address_space(3))) uint32_t *pipe_word_p;
	uint32_t q_n = 0;                         // records this wavefront has pushed (front) / popped (back)
	auto pipe_slot_wait = [&]() {             // front: a free slot
		const pipe_word_p tail = (pipe_word_p)&L.q_tail;
		PIPE_W0(t_wait);
		lat_wait([&]() { return q_n - uniform(*tail) >= INF_PQ; }, 0);
		PIPE_W1(t_wait);                      // (stats build: the front's cycles waiting for a free record)
	
	};
	auto pipe_publish = [&]() {
		asm volatile("" ::: "memory");
		q_n++;
		if (lane == 0)
			*(pipe_word_p)&L.q_head = q_n;
		asm volatile("" ::: "memory");
	
	};
	auto pipe_push_small = [&](uint32_t type, uint32_t x, uint32_t y) {   // a literal, a match, a stored block, the end
		pipe_slot_wait();
		if (lane == 0) {
			uint32_t *h = L.q_hdr[q_n % INF_PQ];
			h[0] = type;
			h[1] = x;
			h[2] = y;
		}
		pipe_publish();
	
	};
	auto pipe_push_window = [&](const WinRec &W) {
		pipe_slot_wait();
		const uint32_t k = q_n % INF_PQ;
		L.q_lane[k][0][lane] = W.lo0;
		L.q_lane[k][1][lane] = W.lo1;
		L.q_lane[k][2][lane] = (W.rel0 & 0xffffu) | (W.rel1 << 16);      // (rel < 1024 on a real token; any other lane holds anything)
		{
			// the header in ONE store: lane j holds word j (eleven selects; a store per word from lane 0 was a sixth of the
			// front wavefront's time)
			const uint32_t hw[11] = { PIPE_WINDOW, W.pos, W.cum, (uint32_t)W.simple0, (uint32_t)(W.simple0 >> 32), (uint32_t)W.simple1,
						  (uint32_t)(W.simple1 >> 32), (uint32_t)W.g0, (uint32_t)(W.g0 >> 32), (uint32_t)W.g1, (uint32_t)(W.g1 >> 32) };
			uint32_t v = 0;
#pragma unroll
			for (uint32_t j = 0; j < 11; j++)
				v = lane == j ? hw[j] : v;
			if (lane < 11)
				L.q_hdr[k][lane] = v;
		}
		pipe_publish();
	};


	// ---- the end of a stream: what is left in the ring, the CRC-32, the verdict (PIPE: the back wavefront's) --------------------
	auto finish = [&](int32_t st) {
		uint32_t crcv = 0;
		if (st == HD_OK) {
			// tail: bytes [flushed, pos) leave the ring byte-wise
			for (uint32_t i = flushed + lane; i < pos; i += 64)
				dst[i] = L.ring[i & (RING - 1)];
			if (want_crc) {
				// full 16-byte slots of the tail piece, then the < 16 byte remainder
				const uint32_t piece = flushed / HD_PIECE;
				const uint32_t o = flushed + 16 * lane;
				const bool full = o + 16 <= pos;
				uint4 v = make_uint4(0, 0, 0, 0);
				if (full)
					v = *(const uint4 *)&L.ring[o & (RING - 1)];
				if (pos < 16)
					crc.s = 0;                       // no full slot at all: finish() reseeds
				crc.fold(ct, piece, full, v);
				crcv = crc.finish(ct, lane, pos, &L.ring[(pos & ~15u) & (RING - 1)]);
			}
		}
		if (lane == 0) {
			a.out_len[b] = st == HD_OK ? pos : 0;
			if (a.status) a.status[b] = st;
			if (a.crc) a.crc[b] = crcv;
		}
	};

	{
		if (back) {
			// ================= the back wavefront: records in order, until the end ==========================================
			const pipe_word_p head = (pipe_word_p)&L.q_head;
			for (;;) {
				PIPE_W0(t_wait);
				lat_wait([&]() { return uniform(*head) == q_n; }, 0);
				PIPE_W1(t_wait);                  // (stats build: the back's cycles waiting for a record)
				if (uniform(*(lat_word_p)&L.abort)) {
					finish(HD_BAD_DATA);              // (a wait did not end: see INF_LAT_SPINS)
					return;
				}
				asm volatile("" ::: "memory");
				const uint32_t k = q_n % INF_PQ;
				const uint32_t type = uniform(L.q_hdr[k][0]), x = uniform(L.q_hdr[k][1]), y = uniform(L.q_hdr[k][2]);
				WinRec W;
				if (type == PIPE_WINDOW) {
					W.lo0 = L.q_lane[k][0][lane];
					W.lo1 = L.q_lane[k][1][lane];
					const uint32_t rl = L.q_lane[k][2][lane];
					W.rel0 = rl & 0xffffu;
					W.rel1 = rl >> 16;
					W.pos = x;
					W.cum = y;
					// (the header's words in one load: word j in lane j)
					const uint32_t hv = L.q_hdr[k][lane < 11 ? lane : 0];
					W.simple0 = ((uint64_t)readlane(hv, 4) << 32) | readlane(hv, 3);
					W.simple1 = ((uint64_t)readlane(hv, 6) << 32) | readlane(hv, 5);
					W.g0 = ((uint64_t)readlane(hv, 8) << 32) | readlane(hv, 7);
					W.g1 = ((uint64_t)readlane(hv, 10) << 32) | readlane(hv, 9);
				}
				// (the record is in registers: its slot is the front's again)
				asm volatile("" ::: "memory");
				q_n++;
				if (lane == 0)
					*(pipe_word_p)&L.q_tail = q_n;
				asm volatile("" ::: "memory");
				if (type == PIPE_WINDOW) {
					place_window(W);
					pos += W.cum;
				} else if (type == PIPE_LITERAL) {
					L.ring[pos & (RING - 1)] = (uint8_t)x;          // (every lane the same byte to the same address)
					pos++;
				} else if (type == PIPE_MATCH) {
					copy_match(x, y);
					pos += x;
				} else if (type == PIPE_STORED) {
					copy_stored(x, y);
				} else {
					finish((int32_t)x);
					PIPE_WFLUSH();
					return;
				}
				if (pos - flushed >= HD_PIECE)
					flush_pieces();
			}
		}
	}

	// ---- PIPE, round 5 (second half): the SPEC wavefront.  What a window's lanes decode -- the token that WOULD start at each bit
	// -- depends on the stream and the tables only, not on where the real chain runs: a third wavefront decodes EVERY bit
	// position of the stream, in chunks of 64 aligned to nothing but the stream itself, and leaves one word per position in an
	// LDS ring { bits of the token | stop flag << 6 | length or literal << 7 | distance << 16 } (distance 0: not a match).  The
	// front reads the 128 words under its window and walks; its two table gathers and ~60 vector instructions per window -- a
	// third of the front's time per window (profiles/r05_inflate_pipe_stats.txt) -- run beside it, ahead of it.
	// The tables are the front's (it parses the headers): sp_stop = E asks the spec wavefront to halt (it answers sp_ack = E
	// and touches nothing), sp_go = E with sp_start = the first chunk restarts it on the new tables; 0xffffffff ends it.
	typedef volatile __attribute__((address_space(3))) uint32_t *spec_word_p;
	if (role == 2) {
		const spec_word_p stopw = (spec_word_p)&L.sp_stop, gow = (spec_word_p)&L.sp_go, tailw = (spec_word_p)&L.sp_tail;
		uint32_t epoch = 0, c = 0;
		uint32_t s_p0 = 0xfffffff0u, s_pre = 0, s_pre_idx = 0xfffffff0u;
		const uint32_t c_end = (((mis + n) >> 2) >> 1) + 2;           // chunks from here on lie behind the stream's last whole dwords
		for (;;) {
			const uint32_t want = uniform(*stopw);
			if (want != epoch) {
				if (want == 0xffffffffu)
					return;
				if (lane == 0)
					*(spec_word_p)&L.sp_ack = want;
				uint32_t g;
				while ((g = uniform(*gow)) != want) {
					if (uniform(*stopw) == 0xffffffffu || uniform(*(lat_word_p)&L.abort))
						return;
					__builtin_amdgcn_s_sleep(1);
				}
				asm volatile("" ::: "memory");
				epoch = want;
				c = uniform(*(spec_word_p)&L.sp_start);
				s_p0 = 0xfffffff0u;
				continue;
			}
			// the ring holds INF_SPEC_CHUNKS chunks from the front's (its window reaches into the two behind its own)
			if (c + 2 > uniform(*tailw) + (INF_SPEC_CHUNKS - 2) || c >= c_end) {
				if (uniform(*(lat_word_p)&L.abort))
					return;
				__builtin_amdgcn_s_sleep(1);
				continue;
			}
			const uint32_t d0 = 2 * c;                      // the pair (c, c + 1): 128 positions from bit 64 c, dword 2 c
			const uint32_t p0 = d0 >> 6;
			if (p0 != s_p0) {
				if (p0 == s_p0 + 1)
					L.comp[lane] = L.comp[64 + lane];
				else
					L.comp[lane] = load_piece(p0);
				L.comp[64 + lane] = s_pre_idx == p0 + 1 ? s_pre : load_piece(p0 + 1);
				s_pre = load_piece(p0 + 2);
				s_pre_idx = p0 + 2;
				s_p0 = p0;
			}
			const uint32_t *wsp = &L.comp[(d0 & 63) + (lane >> 5)];
			const uint32_t ws0 = wsp[0], ws1 = wsp[1], ws2 = wsp[2], ws3 = wsp[3], ws4 = wsp[4];
			auto spec_word = [&](uint32_t bl, uint32_t lo, uint32_t mid, uint32_t hi) -> uint32_t {   // bl = bit offset from dword d0
				const uint32_t a = __builtin_amdgcn_alignbit(mid, lo, bl & 31);
				const uint32_t bq = __builtin_amdgcn_alignbit(hi, mid, bl & 31);
				const uint32_t e = L.lit[a & ((1u << INF_LT_BITS) - 1)];
				const uint32_t len1 = e & 15, eb = (e >> 4) & 15;
				// (a literal's entry has no extra bits: the sum is the literal then)
				const uint32_t lenlit = (e >> 16) + __builtin_amdgcn_ubfe(a, len1, eb);
				const uint32_t t1 = len1 + eb;                 // <= 9 + 5
				const uint32_t rest = __builtin_amdgcn_alignbit(bq, a, t1);
				const uint32_t dd = L.off[rest & ((1u << INF_DT_BITS) - 1)];
				const uint32_t ddm = (e & 0x300) == (K_LEN << 8) ? dd : 0u;
				const uint32_t dlen = ddm & 15, deb = (ddm >> 4) & 15;
				const uint32_t offset = (ddm >> 16) + __builtin_amdgcn_ubfe(rest, dlen, deb);
				const uint32_t tokbits = t1 + dlen + deb;
				// bit 6 = the walk stops in front of this token: bit 9 of either entry (K_EOB and K_SLOW have it) moved down
				const uint32_t tb1 = tokbits ? tokbits : 1u;
				return tb1 | ((e >> 3) & 64u) | ((ddm >> 3) & 64u) | ((lenlit & 511u) << 7) | (offset << 16);
			};
			const uint32_t w0 = spec_word(lane, ws0, ws1, ws2), w1 = spec_word(lane + 64, ws2, ws3, ws4);
			L.spec[(64 * c + lane) & (INF_SPEC_POS - 1)] = w0;
			L.spec[(64 * c + 64 + lane) & (INF_SPEC_POS - 1)] = w1;
			asm volatile("" ::: "memory");
			c += 2;
			if (lane == 0)
				*(spec_word_p)&L.sp_head = c;
			asm volatile("" ::: "memory");
		}
	}
	// the front's side of it
	uint32_t fq_n = 0;                        // records the front has pushed to / the sort wavefront has popped from the queue between them (below)
	uint32_t sp_epoch = 0;
	auto spec_halt = [&]() {
		// (the sort wavefront reads the OLD block's last words from the ring: it has to be through them before the spec
		// wavefront is restarted on the chunk under the new block's first token, which may hold them)
		lat_wait([&]() { return uniform(*(spec_word_p)&L.fq_tail) != fq_n; }, 1);
		sp_epoch++;
		if (lane == 0)
			*(spec_word_p)&L.sp_stop = sp_epoch;
		lat_wait([&]() { return uniform(*(spec_word_p)&L.sp_ack) != sp_epoch; }, 1);
		asm volatile("" ::: "memory");
	};
	auto spec_go = [&]() {
		const uint32_t c0 = (((dw << 5) - bc) >> 6) & ~1u;           // the (even) chunk under the block's first token
		asm volatile("" ::: "memory");
		if (lane == 0) {
			*(spec_word_p)&L.sp_start = c0;
			*(spec_word_p)&L.sp_head = c0;
			*(spec_word_p)&L.sp_tail = c0;
			*(spec_word_p)&L.sp_go = sp_epoch;
		}
		asm volatile("" ::: "memory");
	};
	auto spec_end = [&]() {
		if (lane == 0)
			*(spec_word_p)&L.sp_stop = 0xffffffffu;
	};

	// ---- PIPE, round 5 (third part): the SORT wavefront.  Of what the front did per window after the walk -- prefix sum, budget,
	// distance check, the literals' stores, the matches' classes, the record for the back -- only the first three decide where
	// the NEXT window starts and what the verdict is.  The rest goes to a fourth wavefront: the front hands it { B, the masks of
	// the tokens the walk took, the window's first output byte and its length }, it reads the same words of the spec ring (so
	// the ring's tail is ITS position now), repeats the prefix sum, stores the literals, sorts the matches and pushes the record
	// the back wavefront copies from.  Scalar-path records (a literal, a match, a stored block, the end) pass through it in order.
	auto fq_push = [&](uint32_t type, uint32_t x, uint32_t y, uint32_t Bw, uint64_t r0, uint64_t r1) {
		const pipe_word_p tail = (pipe_word_p)&L.fq_tail;
		PIPE_W0(t_wait);
		lat_wait([&]() { return fq_n - uniform(*tail) >= INF_FQ; }, 0);
		PIPE_W1(t_wait);
		const uint32_t hw[8] = { type, x, y, Bw, (uint32_t)r0, (uint32_t)(r0 >> 32), (uint32_t)r1, (uint32_t)(r1 >> 32) };
		uint32_t v = 0;
#pragma unroll
		for (uint32_t j = 0; j < 8; j++)
			v = lane == j ? hw[j] : v;
		if (lane < 8)
			L.fq_hdr[fq_n % INF_FQ][lane] = v;
		asm volatile("" ::: "memory");
		fq_n++;
		if (lane == 0)
			*(pipe_word_p)&L.fq_head = fq_n;
		asm volatile("" ::: "memory");
	};
	if (role == 3) {
		const pipe_word_p head = (pipe_word_p)&L.fq_head;
		for (;;) {
			lat_wait([&]() { return uniform(*head) == fq_n; }, 0);
			if (uniform(*(lat_word_p)&L.abort))
				return;
			asm volatile("" ::: "memory");
			const uint32_t hv = L.fq_hdr[fq_n % INF_FQ][lane & 7];
			const uint32_t type = readlane(hv, 0), x = readlane(hv, 1), y = readlane(hv, 2), Bw = readlane(hv, 3);
			const uint64_t real0 = ((uint64_t)readlane(hv, 5) << 32) | readlane(hv, 4), real1 = ((uint64_t)readlane(hv, 7) << 32) | readlane(hv, 6);
			asm volatile("" ::: "memory");
			fq_n++;
			if (type != PIPE_WINDOW) {
				if (lane == 0)
					*(pipe_word_p)&L.fq_tail = fq_n;
				asm volatile("" ::: "memory");
				pipe_push_small(type, x, y);
				if (type == PIPE_END)
					return;
				continue;
			}
			const uint32_t wpos = x, cum = y;
			const uint32_t w0 = L.spec[(Bw + lane) & (INF_SPEC_POS - 1)], w1 = L.spec[(Bw + 64 + lane) & (INF_SPEC_POS - 1)];
			asm volatile("" ::: "memory");
			// (the words are in registers: the chunks in front of the window's are the spec wavefront's again -- and only now is
			// the record "popped": the front waits for that before it lets the spec wavefront restart on new tables)
			if (lane == 0) {
				*(spec_word_p)&L.sp_tail = Bw >> 6;
				*(pipe_word_p)&L.fq_tail = fq_n;
			}
			asm volatile("" ::: "memory");
			const uint32_t len0 = (w0 >> 7) & 511u, off0 = w0 >> 16, len1 = (w1 >> 7) & 511u, off1 = w1 >> 16;
			const uint64_t is_len0 = __ballot(off0 != 0), is_len1 = __ballot(off1 != 0);
			const uint32_t outlen0 = sel(is_len0, len0, 1u), outlen1 = sel(is_len1, len1, 1u);
			const uint32_t scn = wave_incl_scan(sel(real0, outlen0, 0u) | (sel(real1, outlen1, 0u) << 16));
			const uint32_t tot = readlane(scn, 63);
			const uint32_t rel0 = (scn & 0xffff) - outlen0, rel1 = (scn >> 16) + (tot & 0xffff) - outlen1;   // valid on the real tokens
			const uint32_t opos0 = wpos + rel0, opos1 = wpos + rel1;
			const uint64_t match0 = real0 & is_len0, match1 = real1 & is_len1;
			// the literals: they depend on nothing, and the ring is nobody's at these bytes until the record is out
			// (lanes without a literal write to their dump slot: no exec juggling, no skip branches)
			const uint64_t lit0 = real0 & ~is_len0, lit1 = real1 & ~is_len1;
			L.ring[sel(lit0, opos0 & (RING - 1), RING + lane)] = (uint8_t)len0;
			L.ring[sel(lit1, opos1 & (RING - 1), RING + lane)] = (uint8_t)len1;
			// the matches, by the way the back wavefront copies them (hd_inflate.hpp's classes without the ring test:
			// this ring holds every source).  "simple": source wholly in front of this window's output, at most 64 bytes
			const uint64_t simple0 = match0 & __ballot(off0 >= rel0 + len0) & __ballot(len0 <= 64);
			const uint64_t simple1 = match1 & __ballot(off1 >= rel1 + len1) & __ballot(len1 <= 64);
			WinRec W;
			W.lo0 = len0 | (off0 << 16);
			W.lo1 = len1 | (off1 << 16);
			W.rel0 = rel0; W.rel1 = rel1; W.pos = wpos; W.cum = cum;
			W.simple0 = simple0; W.simple1 = simple1;
			W.g0 = match0 & ~simple0; W.g1 = match1 & ~simple1;
			pipe_push_window(W);
		}
	}

	auto run_windows = [&](int32_t &st_out) -> uint32_t {
		uint32_t B = (dw << 5) - bc;              // absolute bit position from src32
		uint32_t result = 0;
		for (;;) {
			// (flushing the ring is the back wavefront's; here pos is a number)
			const uint32_t d0 = B >> 5;
			// the budget shrinks to the room that is left, so windows run up to the last bytes of the
			// output (a token that does not fit is cut below and meets the scalar loop's checks)
			const uint32_t budget = cap - pos < WIN_OUT_BUDGET ? cap - pos : WIN_OUT_BUDGET;
			if (!(d0 + 7 <= dw_safe && budget != 0))
				break;
			// the tokens that would start at the window's 128 bit positions: the spec wavefront's words (above)
			PIPE_P0();
			{
				const spec_word_p headw = (spec_word_p)&L.sp_head;
				const uint32_t need = (B + 127) >> 6;
				for (uint32_t spins = 0; uniform(*headw) <= need; spins++) {
					if ((spins & 255u) == 255u && (spins > INF_LAT_SPINS || uniform(*(lat_word_p)&L.abort))) {
						if (lane == 0)
							*(lat_word_p)&L.abort = 1;
						break;
					}
					// The ring's tail is where the sort wavefront reads -- but a run of scalar-path tokens (long codewords,
					// one after the other) moves B on with no window for it to read, and the spec wavefront, held back by
					// a tail some thirty chunks behind B, would never get to the chunk this window needs.  With the sort
					// wavefront through everything handed to it, nobody reads in front of B: the tail is ours to move.
					// (looked at every 16th poll: an ordinary wait here is a few polls long)
					if ((spins & 15u) == 15u && uniform(*(spec_word_p)&L.fq_tail) == fq_n && lane == 0)
						*(spec_word_p)&L.sp_tail = B >> 6;
					__builtin_amdgcn_s_sleep(0);
				}
				asm volatile("" ::: "memory");
			}
			struct Spec {
				uint32_t length, offset, outlen, walk;   // length: a match's, or a literal's byte
				uint64_t is_len, is_lit;             // lane masks
			};
			Spec s0, s1;
			{
				const uint32_t w0 = L.spec[(B + lane) & (INF_SPEC_POS - 1)], w1 = L.spec[(B + 64 + lane) & (INF_SPEC_POS - 1)];
				asm volatile("" ::: "memory");
				// (the ring's tail is the sort wavefront's position: it reads these words after us)
				s0.walk = w0 & 127u; s0.length = (w0 >> 7) & 511u; s0.offset = w0 >> 16;
				s1.walk = w1 & 127u; s1.length = (w1 >> 7) & 511u; s1.offset = w1 >> 16;
				s0.is_len = __ballot(s0.offset != 0); s0.is_lit = ~s0.is_len;    // (of the tokens the walk takes: it stops in front of the others)
				s1.is_len = __ballot(s1.offset != 0); s1.is_lit = ~s1.is_len;
				s0.outlen = sel(s0.is_len, s0.length, 1u);
				s1.outlen = sel(s1.is_len, s1.length, 1u);
			}

			// The real chain from bit 0 of the window.  This walk is the hottest scalar
			// code of the kernel (the CU has one scalar ALU) and the compiler spends ~20
			// instructions per token on it, so it is written out: 4 SALU + 2 branches
			// + 1 v_readlane per token (the lane select of v_readlane and s_bitset1 take
			// the low 6 bits, so the second half runs on b itself).  It stops in front of
			// the first token the window cannot take; that one goes to the scalar loop.
			// (A lane select written by the SALU needs no wait states before v_readlane,
			// only one written by the VALU does.)
			PIPE_P(0);                                   // the wait for the spec wavefront's words, their fields
			uint32_t b, wm;
			uint64_t real0, real1;
			// (round 5: the stop flag is bit 6 of the word that is ADDED to the position -- a token the walk must stop in front of
			// throws it out of the half by itself, and the half's loop needs no test of its own for it: five instructions and one
			// branch per token instead of seven and two; behind the loop the last word says whether its token has to be taken back)
			asm volatile("s_mov_b32 %0, 0\n\t"
				     "s_mov_b64 %1, 0\n\t"
				     "s_mov_b64 %2, 0\n"
				     "Lhd_walk0_%=:\n\t"
				     "v_readlane_b32 %3, %4, %0\n\t"
				     "s_bitset1_b64 %1, %0\n\t"
				     "s_add_u32 %0, %0, %3\n\t"
				     "s_cmp_lt_u32 %0, 64\n\t"
				     "s_cbranch_scc1 Lhd_walk0_%=\n\t"
				     "s_bitcmp1_b32 %3, 6\n\t"
				     "s_cbranch_scc0 Lhd_walk1_%=\n\t"
				     "s_sub_u32 %0, %0, %3\n\t"
				     "s_bitset0_b64 %1, %0\n\t"
				     "s_branch Lhd_walk_done_%=\n"
				     "Lhd_walk1_%=:\n\t"
				     "v_readlane_b32 %3, %5, %0\n\t"
				     "s_bitset1_b64 %2, %0\n\t"
				     "s_add_u32 %0, %0, %3\n\t"
				     "s_cmp_lt_u32 %0, 128\n\t"
				     "s_cbranch_scc1 Lhd_walk1_%=\n\t"
				     "s_bitcmp1_b32 %3, 6\n\t"
				     "s_cbranch_scc0 Lhd_walk_done_%=\n\t"
				     "s_sub_u32 %0, %0, %3\n\t"
				     "s_bitset0_b64 %2, %0\n"
				     "Lhd_walk_done_%=:"
				     : "=&s"(b), "=&s"(real0), "=&s"(real1), "=&s"(wm)
				     : "v"(s0.walk), "v"(s1.walk)
				     : "scc");
			// output positions; cut in front of the first token that would overrun the budget.  (Conditions are
			// 64-bit lane masks: the ballot of ONE compare each, combined in scalar code, back to the lanes
			// through sel() -- hd_device.hpp "lane masks".)
			// (both halves' output lengths in one prefix sum, 16 bits each: 64 x 258 < 2^16)
			PIPE_P(1);                                   // the walk
			const uint32_t scn = wave_incl_scan(sel(real0, s0.outlen, 0u) | (sel(real1, s1.outlen, 0u) << 16));
			const uint32_t tot = readlane(scn, 63);                   // the totals of both halves
			const uint32_t incl0 = scn & 0xffff;
			const uint32_t incl1 = (scn >> 16) + (tot & 0xffff);
			// (the usual window fits its budget whole: one scalar compare, and the total is its output)
			uint32_t cum = (tot & 0xffff) + (tot >> 16);
			const bool over = cum > budget;
			if (over) {
				const uint64_t over0 = __ballot(incl0 > budget) & real0;
				const uint64_t over1 = __ballot(incl1 > budget) & real1;
				if (over0) {
					b = (uint32_t)__ffsll((unsigned long long)over0) - 1;
					real0 &= (1ull << b) - 1;
					real1 = 0;
				} else {
					const uint32_t f = (uint32_t)__ffsll((unsigned long long)over1) - 1;
					real1 &= (1ull << f) - 1;
					b = 64 + f;
				}
				if (real0)
					cum = real1 ? readlane(incl1, 63 - (uint32_t)__clzll((long long)real1))
						    : readlane(incl0, 63 - (uint32_t)__clzll((long long)real0));
			}
			PIPE_P(2);                                   // prefix sum, budget
			if (real0 == 0) {
				break;                                     // the token at B is not for a window: scalar loop
			}
			{
				// offset > bytes out so far: decompress_template.h:724.  (No distance is longer than 32768: once that much is
				// out the test cannot fire, and half the windows of a 64 KiB block skip its dozen instructions)
				if (pos < 32768u) {
					const uint32_t rel0 = incl0 - s0.outlen, rel1 = incl1 - s1.outlen;   // valid on the real tokens
					const uint32_t opos0 = pos + rel0, opos1 = pos + rel1;
					const uint64_t match0 = real0 & s0.is_len, match1 = real1 & s1.is_len;
					uint64_t far0, far1;
					asm("v_cmp_lt_u32 %0, %1, %2" : "=s"(far0) : "v"(opos0), "v"(s0.offset));
					asm("v_cmp_lt_u32 %0, %1, %2" : "=s"(far1) : "v"(opos1), "v"(s1.offset));
					if ((far0 & match0) | (far1 & match1)) {
						st_out = HD_BAD_DATA;
						result = 2;
						break;
					}
				}
				PIPE_P(3);                                   // the distance check
				// the literals' stores, the matches' classes and the record are the sort wavefront's (above)
				fq_push(PIPE_WINDOW, pos, cum, B, real0, real1);
				PIPE_P(4);                                   // the hand-over (and the wait for its slot)
			}
			pos += cum;
			B += b;
			// the walk stopped in front of a token no window takes (long codeword, end of block): a new
			// window there would come back empty (6 % of all windows did) -- the scalar loop is next
			if ((wm & 64) && !over)
				break;
		}
		// hand the position back to the scalar reader
		dw = B >> 5;
		if ((dw >> 6) != cur_piece) {
			const uint32_t piece = dw >> 6;
			if (piece == cur_piece + 1)
				cw = cw_next;
			else
				cw = load_piece(piece);
			cur_piece = piece;
			cw_next = load_piece(piece + 1);
		}
		bb = 0;
		bc = 0;
		refill();
		bb >>= (B & 31);
		bc -= (B & 31);
		return result;
	};

	int32_t st = HD_OK;
	bool static_loaded = false;
	// The direct tables are built in LDS, then kept in VGPRs for the symbol loop:
	// entry i lives in lane i & 63 of register i >> 6, a lookup is one relative
	// v_mov (s_set_gpr_idx) + v_readlane -- no LDS round trip per symbol.
	typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
	u32x8 LT;
	auto load_tables = [&]() {
#pragma unroll
		for (int r = 0; r < (1 << INF_LT_BITS) / 64; r++)
			LT[r] = L.lit[r * 64 + lane];
	};

	for (;;) {
		refill();
		const uint32_t bfinal = (uint32_t)bb & 1;
		const uint32_t btype = ((uint32_t)bb >> 1) & 3;
		bb >>= 3;
		bc -= 3;

		if (btype == 0) {
			// ---- stored: decompress_template.h:234-279 -------------------
			int64_t cbits = (consumed_bits() + 7) & ~(int64_t)7;
			if (cbits > 8 * (int64_t)n) { st = HD_BAD_DATA; break; }
			uint32_t ip = (uint32_t)(cbits >> 3);
			if (n - ip < 4) { st = HD_BAD_DATA; break; }
			seek_byte(ip);
			refill();
			const uint32_t len = (uint32_t)bb & 0xffff, nlen = ((uint32_t)bb >> 16) & 0xffff;
			ip += 4;
			if (len != (~nlen & 0xffff)) { st = HD_BAD_DATA; break; }
			if (len > cap - pos) { st = HD_INSUFFICIENT_SPACE; break; }
			if (len > n - ip) { st = HD_BAD_DATA; break; }
			fq_push(PIPE_STORED, ip, len, 0, 0, 0);
			pos += len;
			seek_byte(ip + len);
		} else if (btype == 3) {
			st = HD_BAD_DATA;
			break;
		} else {
			uint32_t nlit = 288, noff = 32;
			spec_halt();                              // the tables are about to change: the spec wavefront stands still
			if (btype == 2) {
				// ---- dynamic header: decompress_template.h:101-232 -------
				refill();
				nlit = 257 + ((uint32_t)bb & 31);
				noff = 1 + (((uint32_t)bb >> 5) & 31);
				const uint32_t npre = 4 + (((uint32_t)bb >> 10) & 15);
				bb >>= 14;
				bc -= 14;
				if (lane < 19)
					L.pre_lens[lane] = 0;
				for (uint32_t i = 0; i < npre; i++) {
					refill();
					if (lane == 0)
						L.pre_lens[k_precode_perm[i]] = (uint8_t)((uint32_t)bb & 7);
					bb >>= 3;
					bc -= 3;
				}
				if (!uniform(build_table<2>(L.pre_lens, 19, L.lit, 7, L.lit_sorted, nullptr, L.cl + INF_T_SCRATCH, lane))) { st = HD_BAD_DATA; break; }
				uint8_t *cl = L.cl;
				uint32_t i = 0, prev = 0;
				bool bad = false;
				while (i < nlit + noff) {
					refill();
					if (consumed_bits() > 8 * (int64_t)n + 64) { bad = true; break; }
					const uint32_t e = uniform(L.lit[(uint32_t)bb & 127]);
					const uint32_t cl_len = e & 15, s = e >> 16;
					bb >>= cl_len;
					bc -= cl_len;
					if (s < 16) {
						if (lane == 0)
							cl[i] = (uint8_t)s;
						prev = s;
						i++;
						continue;
					}
					uint32_t rep, val = 0;
					if (s == 16) {
						if (i == 0) { bad = true; break; }
						rep = 3 + ((uint32_t)bb & 3);
						bb >>= 2; bc -= 2;
						val = prev;
					} else if (s == 17) {
						rep = 3 + ((uint32_t)bb & 7);
						bb >>= 3; bc -= 3;
						prev = 0;
					} else {
						rep = 11 + ((uint32_t)bb & 127);
						bb >>= 7; bc -= 7;
						prev = 0;
					}
					for (uint32_t k = lane; k < rep; k += 64)
						cl[i + k] = (uint8_t)val;
					i += rep;
				}
				if (bad || i != nlit + noff) { st = HD_BAD_DATA; break; }
				static_loaded = false;
				if (!uniform(build_table<1>(cl + nlit, noff, L.off, INF_DT_BITS, L.off_sorted, L.off_count, L.cl + INF_T_SCRATCH, lane)) ||
				    !uniform(build_table<0>(cl, nlit, L.lit, INF_LT_BITS, L.lit_sorted, L.lit_count, L.cl + INF_T_SCRATCH, lane))) {
					st = HD_BAD_DATA;
					break;
				}
			} else if (!static_loaded) {
				// ---- static code: decompress_template.h:297-330 ----------
				uint8_t *cl = L.cl;
				for (uint32_t s = lane; s < 320; s += 64)
					cl[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : s < 288 ? 8 : 5;
				build_table<1>(cl + 288, 32, L.off, INF_DT_BITS, L.off_sorted, L.off_count, L.cl + INF_T_SCRATCH, lane);
				build_table<0>(cl, 288, L.lit, INF_LT_BITS, L.lit_sorted, L.lit_count, L.cl + INF_T_SCRATCH, lane);
				static_loaded = true;
			}

			load_tables();
			spec_go();                                // ... and starts on the new ones at the block's first token
			// ---- symbol loop ----------------------------------------------
			for (;;) {
				{
					const uint32_t wr = uniform(run_windows(st));
					if (wr)                      // 1: end of block consumed, 2: error
						break;
				}
				// one token through the fully checked scalar path (stream edges, long codes)
				// one flush site for the whole symbol loop (at most 1023 + 258 bytes pending)
				refill();
				if (overrun()) { st = HD_BAD_DATA; break; }
				const uint32_t li = (uint32_t)bb & ((1u << INF_LT_BITS) - 1);
				uint32_t e = readlane(LT[li >> 6], li & 63);
				if (((e >> 8) & 3) == K_SLOW) {
					const uint32_t sl = uniform(slow_decode(bb, L.lit_count, L.lit_sorted, lane));
					e = litlen_entry(sl & 0xffff, sl >> 16);
				}
				const uint32_t clen = e & 15;
				bb >>= clen;
				bc -= clen;
				const uint32_t kind = (e >> 8) & 3;
				if (kind == K_LIT) {
					if (pos == cap) { st = HD_INSUFFICIENT_SPACE; break; }
					fq_push(PIPE_LITERAL, e >> 16, 0, 0, 0, 0);
					pos++;
					continue;
				}
				if (kind == K_EOB) {
					break;
				}
				const uint32_t eb = (e >> 4) & 15;
				const uint32_t length = (e >> 16) + ((uint32_t)bb & ((1u << eb) - 1));
				bb >>= eb;
				bc -= eb;
				if (length > cap - pos) { st = HD_INSUFFICIENT_SPACE; break; }
				refill();
				const uint32_t di = (uint32_t)bb & ((1u << INF_DT_BITS) - 1);
				uint32_t d = uniform(L.off[di]);                 // (LDS: one token in 130 comes this way, its copy in four registers cost a wave)
				if (((d >> 8) & 3) == K_SLOW) {
					const uint32_t sl = uniform(slow_decode(bb, L.off_count, L.off_sorted, lane));
					d = offset_entry(sl & 0xffff, sl >> 16);
				}
				const uint32_t dlen = d & 15, deb = (d >> 4) & 15;
				bb >>= dlen;
				bc -= dlen;
				const uint32_t offset = (d >> 16) + ((uint32_t)bb & ((1u << deb) - 1));
				bb >>= deb;
				bc -= deb;
				if (offset > pos) { st = HD_BAD_DATA; break; }

				fq_push(PIPE_MATCH, length, offset, 0, 0, 0);
				pos += length;
			}
			if (st != HD_OK)
				break;
		}
		if (bfinal)
			break;
		if ((a.flags & INF_FLUSHED) && consumed_bits() <= 8 * (int64_t)n && consumed_bits() + 7 >= 8 * (int64_t)n)
			break;
		if (consumed_bits() > 8 * (int64_t)n + 64) { st = HD_BAD_DATA; break; }
	}
	if (st == HD_OK && consumed_bits() > 8 * (int64_t)n)
		st = HD_BAD_DATA;

	spec_end();
	fq_push(PIPE_END, (uint32_t)st, 0, 0, 0, 0);                // (through the sort wavefront) the back wavefront writes the tail and the verdict
	PIPE_WFLUSH();
}

// latency form (hip_inflate / hip_inflate_flush: a handful of streams, callers waiting): the same decoder with the whole
// DEFLATE window -- a whole BGZF block -- in LDS.  One wavefront alone on its CU cannot hide a load of flushed output
// behind other waves; here it never issues one (a match reaches back 32 KiB at most, the ring holds 64).
__global__ __launch_bounds__(INF_LAT_THREADS) void k_inflate_lat(InflateArgs a)
{
	__shared__ InfLdsPipeT<INF_RING_LAT> L;
	inflate_stream_pipe<INF_RING_LAT, InfLdsPipeT<INF_RING_LAT>>(a, L);
}

} // namespace hd
