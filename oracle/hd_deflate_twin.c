/*
 * hd_deflate_twin.c -- oracle: serial CPU twin of the GPU encoder.
 * TEST INFRASTRUCTURE ONLY (see hd_oracle.h).
 *
 * What it restates.  The reference's encoder for this path is libdeflate
 * (lib/libdeflate/deflate_compress.c:2453-2524 greedy/ht at level 1,
 * :2606-2809 lazy/hc at level 6, :1707-2038 block flush).  libdeflate's
 * compressed bytes are explicitly not a stable golden
 * (lib/libdeflate/libdeflate.h:75-83), so the GPU encoder is free to parse
 * differently as long as the result is valid RFC 1951 that inflates to the
 * input.  It does parse differently -- 64 positions per step instead of one
 * -- and THIS file is the byte-exact serial statement of that parse, so the
 * kernel has something to be bit-compared with.  The role correspondence:
 *
 *   ht_matchfinder_longest_match (ht_matchfinder.h:78)  -> step lookups below
 *   deflate_choose_match/_literal (deflate_compress.c:2231,2244) -> token list
 *   deflate_flush_block static/stored choice (:1750-1867) -> finish()
 *   length/offset slot tables (:237-318)                -> len_slot()/off_slot()
 *
 * The algorithm (one "step" = what one wavefront does at once):
 *   Steps stand on fixed 64-byte strides: S = 0, 64, 128, ...  Lanes l = 0..63
 *   stand on p = S + l; `carry` = how many leading positions the previous
 *   steps' last match still covers (it may cover whole steps).
 *   1. every lane with 4 bytes left hashes in[p..p+4) and reads the table
 *      entry (latest earlier position with that hash, from PREVIOUS steps only),
 *   2. then every such lane -- covered or not -- publishes p:
 *      table[h] = max(table[h], p + 1),
 *   3. a lane is a match start candidate iff its entry is inside the window and
 *      the 4 bytes there are equal,
 *   4. greedy resolution left to right from lane `carry`: the first candidate at
 *      or after the cursor E is taken, extended to its full length (<= 258,
 *      <= n - p), and E jumps past it; lanes not covered are literals,
 *   5. tokens are emitted in position order; carry = max(E - 64, 0).
 */
#include <stdlib.h>
#include <string.h>
#include "hd_oracle.h"
#include "../include/hipdeflate_params.h"

/* ---- LSB-first bit writer into a zeroed buffer ------------------------- */
typedef struct {
	uint8_t *buf;
	uint64_t bitpos;
} bw_t;

static void bw_put(bw_t *w, uint32_t bits, unsigned n)
{
	for (unsigned i = 0; i < n; i++, w->bitpos++)
		if ((bits >> i) & 1)
			w->buf[w->bitpos >> 3] |= (uint8_t)(1u << (w->bitpos & 7));
}

static uint32_t bitrev(uint32_t v, unsigned n)
{
	uint32_t r = 0;
	for (unsigned i = 0; i < n; i++)
		r |= ((v >> i) & 1) << (n - 1 - i);
	return r;
}

/* ---- RFC 1951 3.2.5 slot arithmetic (tables at deflate_compress.c:237-318) */
static unsigned ilog2(uint32_t v) { unsigned r = 0; while (v >>= 1) r++; return r; }

/* length 3..258 -> litlen symbol, number of extra bits, extra value */
static void len_slot(unsigned len, unsigned *sym, unsigned *ebits, unsigned *eval)
{
	unsigned l = len - 3;
	if (l < 8) { *sym = 257 + l; *ebits = 0; *eval = 0; return; }
	if (l == 255) { *sym = 285; *ebits = 0; *eval = 0; return; }
	unsigned e = ilog2(l) - 2;
	*sym = 261 + 4 * e + ((l >> e) & 3);
	*ebits = e;
	*eval = l & ((1u << e) - 1);
}

/* offset 1..32768 -> offset symbol, number of extra bits, extra value */
static void off_slot(unsigned off, unsigned *sym, unsigned *ebits, unsigned *eval)
{
	unsigned d = off - 1;
	if (d < 4) { *sym = d; *ebits = 0; *eval = 0; return; }
	unsigned e = ilog2(d) - 1;
	*sym = 2 * e + 2 + ((d >> e) & 1);
	*ebits = e;
	*eval = d & ((1u << e) - 1);
}

/* ---- static Huffman code (RFC 1951 3.2.6), codewords sent MSB first ----- */
static void static_litlen(unsigned sym, uint32_t *code, unsigned *n)
{
	if (sym < 144)      { *code = 0x30 + sym;          *n = 8; }
	else if (sym < 256) { *code = 0x190 + (sym - 144); *n = 9; }
	else if (sym < 280) { *code = sym - 256;           *n = 7; }
	else                { *code = 0xC0 + (sym - 280);  *n = 8; }
	*code = bitrev(*code, *n);
}

static void put_static_literal(bw_t *w, unsigned byte)
{
	uint32_t c; unsigned n;
	static_litlen(byte, &c, &n);
	bw_put(w, c, n);
}

static void put_static_match(bw_t *w, unsigned len, unsigned off)
{
	uint32_t c; unsigned n, sym, eb, ev;
	len_slot(len, &sym, &eb, &ev);
	static_litlen(sym, &c, &n);
	bw_put(w, c, n);
	bw_put(w, ev, eb);
	off_slot(off, &sym, &eb, &ev);
	bw_put(w, bitrev(sym, 5), 5);
	bw_put(w, ev, eb);
}

/* ---- one parse step ---------------------------------------------------- */
typedef struct {
	uint16_t *table;      /* HD_TABLE_ENTRIES entries: (position + 1) mod 2^16, 0 = empty */
	unsigned hash_bits;
	unsigned win;         /* ring size in bytes */
	size_t filled;        /* bytes the GPU ring has been filled up to */
	unsigned intra;       /* candidates at distances 1..intra inside the step (HD_INTRA_DIST) */
} mf_t;
/* (rounds 2-4 also had the lazy levels' one-wavefront matchfinder here -- six-byte key, two positions per bucket, a one-lane
 * lazy rule -- for the latency form of levels 3..9; since round 5 those levels are the workgroup parse in every form,
 * deflate_wg() below, and the step parse is level 1's and level 2's: greedy, minimum length 4) */

typedef struct {
	unsigned lanes;           /* positions covered by lanes this step */
	unsigned carry_out;       /* lanes of the NEXT step the last match covers */
	uint8_t  is_match[HD_WAVE];
	uint8_t  is_lit[HD_WAVE];
	uint16_t len[HD_WAVE];
	uint32_t dist[HD_WAVE];
} step_t;

static uint32_t load32(const uint8_t *p)
{
	return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24);
}

/* table slot of the four bytes v (HD_HASH_SLOT, include/hipdeflate_params.h) */
static uint32_t mf_index(const mf_t *mf, uint32_t v)
{
	return HD_HASH_SLOT(v, HD_TABLE_ENTRIES((unsigned)__builtin_ctz(mf->win), mf->hash_bits));
}

static void parse_step(mf_t *mf, const uint8_t *in, size_t n, size_t S, unsigned carry, step_t *st)
{
	uint32_t cand[HD_WAVE];
	uint8_t ok[HD_WAVE];
	unsigned lanes = n - S < HD_WAVE ? (unsigned)(n - S) : HD_WAVE;
	const unsigned keyb = HD_MIN_MATCH;       /* bytes a position needs to be hashed */

	/* the ring is refilled a 1 KiB piece at a time until it holds
	 * HD_LOOKAHEAD bytes past S (or the whole input) */
	while (mf->filled < n && mf->filled < S + HD_LOOKAHEAD)
		mf->filled += HD_PIECE;
	size_t lo = mf->filled > mf->win ? mf->filled - mf->win : 0;

	memset(st, 0, sizeof(*st));
	for (unsigned l = 0; l < lanes; l++) {          /* 1. look up */
		size_t p = S + l;
		ok[l] = 0;
		if (p + keyb > n)
			continue;
		uint32_t v = load32(in + p);
		uint32_t e = mf->table[mf_index(mf, v)];
		/* the latest p' < p with p' + 1 == e (mod 2^16); for inputs <= 64 KiB
		 * that is simply e */
		uint32_t back = (uint32_t)(p + 1 - e) & 0xffffu;   /* 0 = exactly 2^16 back: stale */
		cand[l] = (e && back) ? (uint32_t)(p + 1 - back) : 0;
		/* The table only knows earlier steps, and DNA-like data is full of repeats a few bytes
		 * apart (quality strings).  A lane also looks at the lane(s) just before it in the same
		 * step; the nearest one with the same four bytes is the latest occurrence and replaces
		 * the table's candidate (the kernel compares DPP-shifted copies). */
		for (unsigned d = 1; d <= mf->intra && d <= l; d++)
			if (load32(in + p - d) == v) {
				cand[l] = (uint32_t)(p - d + 1);
				break;
			}
	}
	for (unsigned l = 0; l < lanes; l++) {          /* 2. publish */
		size_t p = S + l;
		if (p + keyb > n)
			continue;
		/* the lanes of a step that hash alike store to one entry in one instruction: the highest lane's
		 * data stays (the device self-test probes it), i.e. within a step the last lane wins */
		mf->table[mf_index(mf, load32(in + p))] = (uint16_t)(p + 1);
	}
	for (unsigned l = 0; l < lanes; l++) {          /* 3. verify */
		size_t p = S + l;
		if (p + keyb > n)
			continue;
		if (cand[l] == 0)
			continue;
		size_t c = cand[l] - 1;
		if (c < lo)
			continue;
		if (load32(in + c) != load32(in + p))
			continue;
		ok[l] = 1;                     /* four equal bytes: a candidate (the minimum match of levels 1 and 2) */
		st->dist[l] = (uint32_t)(p - c);
	}
	unsigned E = carry;                             /* 4. greedy */
	for (unsigned l = 0; l < lanes; l++) {
		if (l < E || !ok[l])
			continue;
		size_t p = S + l;
		unsigned maxlen = n - p < HD_MAX_MATCH ? (unsigned)(n - p) : HD_MAX_MATCH;
		unsigned len = HD_MIN_MATCH;
		while (len < maxlen && in[p + len] == in[p + len - st->dist[l]])
			len++;
		st->is_match[l] = 1;
		st->len[l] = (uint16_t)len;
		E = l + len;
	}
	E = carry;
	for (unsigned l = 0; l < lanes; l++) {
		if (st->is_match[l])
			E = l + st->len[l];
		else if (l >= E)
			st->is_lit[l] = 1;
	}
	st->lanes = lanes;
	st->carry_out = E > lanes ? E - lanes : 0;   /* E >= carry always */
}

static int write_stored(uint8_t *dest, size_t *destLen, const uint8_t *in, size_t n, int flush);
static void put_flush_suffix(bw_t *w);

/* ---- level 1: greedy + static Huffman, streaming ------------------------ */
/* prime: 0, or -- a latency-mode segment behind the first of its block (HD_LAT_PRIME) -- the bytes before `in` whose steps run
 * ahead of the segment's own: they fill table and window, their tokens are dropped, no match crosses the border */
static int deflate_static(uint8_t *dest, size_t *destLen, const uint8_t *in, size_t n,
			  unsigned win_bits, unsigned hash_bits, int flush, size_t prime)
{
	/* flush form: 5 bytes are kept free for the suffix (put_flush_suffix) */
	size_t cap = *destLen >= (flush ? 5u : 0u) ? *destLen - (flush ? 5u : 0u) : 0;
	size_t stored = HD_STORED_SIZE(n);
	/* the static stream is kept only if it ends up strictly smaller than the
	 * stored form and fits; tie -> stored, as deflate_flush_block's
	 * "stored > static > dynamic" preference (deflate_compress.c:1820-1867) */
	size_t limit = cap < stored - 1 ? cap : stored - 1;
	uint8_t *tmp = calloc(1, limit + HD_STEP_MAX_BITS / 8 + 16 + 8);
	mf_t mf = { calloc((size_t)1 << hash_bits, 2), hash_bits, 1u << win_bits, 0, 0 };                /* level 1 is the speed level: no run candidates */
	bw_t w = { tmp, 0 };
	int use_static = 1;
	step_t st;
	uint8_t scratch[HD_STEP_MAX_BITS / 8 + 8];

	bw_put(&w, flush ? 0 : 1, 1);       /* BFINAL */
	bw_put(&w, 1, 2);       /* BTYPE = 01 */
	unsigned carry = 0;
	const uint8_t *const in_own = in;
	const size_t n_own = n;
	in -= prime;
	n += prime;
	for (size_t S = 0; S < n && use_static; S += HD_WAVE) {
		if (S == prime)
			carry = 0;
		parse_step(&mf, in, n, S, carry, &st);
		carry = st.carry_out;
		if (S < prime)
			continue;
		/* the kernel knows the step's bit count (wave prefix sum) before it
		 * writes anything; once the stream plus the end-of-block code can no
		 * longer fit in `limit` bytes it abandons the static stream for good */
		bw_t probe = { scratch, 0 };
		memset(scratch, 0, sizeof(scratch));
		for (unsigned l = 0; l < st.lanes; l++) {
			if (st.is_match[l])
				put_static_match(&probe, st.len[l], st.dist[l]);
			else if (st.is_lit[l])
				put_static_literal(&probe, in[S + l]);
		}
		if (w.bitpos + probe.bitpos + 7 > 8 * (uint64_t)limit) {
			use_static = 0;
			break;
		}
		for (unsigned l = 0; l < st.lanes; l++) {
			if (st.is_match[l])
				put_static_match(&w, st.len[l], st.dist[l]);
			else if (st.is_lit[l])
				put_static_literal(&w, in[S + l]);
		}
	}
	if (use_static && w.bitpos + 7 > 8 * (uint64_t)limit)
		use_static = 0;
	int ret = 0;
	if (use_static) {
		bw_put(&w, 0, 7);  /* end of block */
		if (flush)
			put_flush_suffix(&w);
		*destLen = (size_t)((w.bitpos + 7) >> 3);
		memcpy(dest, tmp, *destLen);
	} else {
		ret = write_stored(dest, destLen, in_own, n_own, flush);
	}
	free(mf.table);
	free(tmp);
	return ret;
}

/* ---- levels >= 2: dynamic Huffman ------------------------------------------
 * Role of deflate_make_huffman_code (deflate_compress.c:1319-1396),
 * deflate_precompute_huffman_header (:1571-1631) and the dynamic branch of
 * deflate_flush_block (:1861-1926), restated with the (deliberately simple,
 * serial) algorithms the kernel runs on one lane:
 *   - symbols sorted by (frequency, symbol) through an explicit rank,
 *   - two-queue Huffman merge, depths from parent links,
 *   - lengths limited by moving overflow up the tree level counts,
 *   - canonical codewords in symbol order,
 *   - greedy RLE of the code lengths with precode symbols 16/17/18. */
typedef struct {
	uint8_t len[288];
	uint16_t code[288];    /* bit-reversed, ready for the LSB-first writer */
} huff_t;

static void build_code(const uint32_t *freq_in, unsigned nsyms, unsigned maxbits, huff_t *h)
{
	uint32_t freq[288];
	uint16_t order[288];
	uint32_t nf[576];
	uint16_t parent[576];
	uint8_t depth[576];
	unsigned blc[16] = { 0 };
	unsigned nu = 0;

	memcpy(freq, freq_in, nsyms * 4);
	for (unsigned s = 0; s < nsyms; s++)
		nu += freq[s] != 0;
	/* at least two codewords (old decoders; deflate_compress.c:1369-1378) */
	if (nu == 0) {
		freq[0] = freq[1] = 1;
	} else if (nu == 1) {
		freq[freq[0] ? 1 : 0] = 1;
	}
	nu = 0;
	for (unsigned s = 0; s < nsyms; s++)
		nu += freq[s] != 0;
	/* rank = number of used symbols that sort before s */
	for (unsigned s = 0; s < nsyms; s++) {
		if (!freq[s])
			continue;
		unsigned r = 0;
		for (unsigned t = 0; t < nsyms; t++)
			if (freq[t] && (freq[t] < freq[s] || (freq[t] == freq[s] && t < s)))
				r++;
		order[r] = (uint16_t)s;
	}
	for (unsigned i = 0; i < nu; i++)
		nf[i] = freq[order[i]];
	/* two-queue merge: leaves 0..nu-1 ascending, internal nodes nu..2nu-2 */
	unsigned i = 0, j = nu, k = nu;
	while (k < 2 * nu - 1) {
		unsigned pick[2];
		for (int t = 0; t < 2; t++) {
			if (i < nu && (j >= k || nf[i] <= nf[j]))
				pick[t] = i++;
			else
				pick[t] = j++;
		}
		nf[k] = nf[pick[0]] + nf[pick[1]];
		parent[pick[0]] = parent[pick[1]] = (uint16_t)k;
		k++;
	}
	depth[2 * nu - 2] = 0;
	for (int x = (int)(2 * nu - 3); x >= 0; x--)
		depth[x] = (uint8_t)(depth[parent[x]] + 1);
	/* level counts: leaves deeper than maxbits are cut back to maxbits, which leaves the code
	 * over-subscribed by `excess` codewords of length maxbits (Kraft sum in units of 2^-maxbits);
	 * every pass below gives one of them back: a leaf moves one level down, a maxbits leaf becomes
	 * its sibling.  (Counting the cut leaves, as deflate's gen_bitlen does after clamping depths
	 * on the way down, is only right for leaves at maxbits + 1: a deeper one frees less.) */
	for (unsigned x = 0; x < nu; x++)
		blc[depth[x] > maxbits ? maxbits : depth[x]]++;
	long excess = -(1L << maxbits);
	for (unsigned bits = 1; bits <= maxbits; bits++)
		excess += (long)blc[bits] << (maxbits - bits);
	while (excess > 0) {
		unsigned bits = maxbits - 1;
		while (bits >= 1 && blc[bits] == 0)
			bits--;
		if (bits == 0)
			break;                  /* only with more symbols than 2^maxbits: no such code exists */
		blc[bits]--;
		blc[bits + 1] += 2;
		blc[maxbits]--;
		excess--;
	}
	memset(h->len, 0, sizeof(h->len));
	unsigned idx = 0;
	for (unsigned bits = maxbits; bits >= 1; bits--)
		for (unsigned c = blc[bits]; c; c--)
			h->len[order[idx++]] = (uint8_t)bits;
	/* canonical codewords */
	unsigned next[17] = { 0 }, code = 0;
	for (unsigned bits = 1; bits <= maxbits; bits++) {
		code = (code + blc[bits - 1]) << 1;
		next[bits] = code;
	}
	blc[0] = 0;
	memset(h->code, 0, sizeof(h->code));
	for (unsigned s = 0; s < nsyms; s++)
		if (h->len[s])
			h->code[s] = (uint16_t)bitrev(next[h->len[s]]++, h->len[s]);
}

static const uint8_t precode_perm[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };

/* RLE items: low 5 bits symbol, bits 8.. extra value */
static unsigned rle_lens(const uint8_t *lens, unsigned total, uint16_t *items, uint32_t *pfreq)
{
	unsigned ni = 0, i = 0;
	while (i < total) {
		unsigned v = lens[i], run = 1;
		while (i + run < total && lens[i + run] == v)
			run++;
		i += run;
		if (v == 0) {
			while (run >= 11) {
				unsigned r = run < 138 ? run : 138;
				items[ni++] = (uint16_t)(18 | ((r - 11) << 8));
				pfreq[18]++;
				run -= r;
			}
			if (run >= 3) {
				items[ni++] = (uint16_t)(17 | ((run - 3) << 8));
				pfreq[17]++;
				run = 0;
			}
		} else {
			items[ni++] = (uint16_t)v;
			pfreq[v]++;
			run--;
			while (run >= 3) {
				unsigned r = run < 6 ? run : 6;
				items[ni++] = (uint16_t)(16 | ((r - 3) << 8));
				pfreq[16]++;
				run -= r;
			}
		}
		while (run--) {
			items[ni++] = (uint16_t)v;
			pfreq[v]++;
		}
	}
	return ni;
}

typedef struct {
	uint32_t lf[288], df[32];
	uint32_t *tok;
	unsigned ntok;
} dynblk_t;

static void static_lens(uint8_t *ll, uint8_t *dl)
{
	for (unsigned s = 0; s < 288; s++)
		ll[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
	for (unsigned s = 0; s < 32; s++)
		dl[s] = 5;
}

/* emit one DEFLATE block from the token slab; returns 0 if it would push the
 * payload past `limit` bytes (the kernel then abandons for the stored form) */
static int flush_dyn_block(bw_t *w, dynblk_t *b, int final, uint64_t limit_bits)
{
	huff_t lh, dh, ph;
	uint16_t items[288 + 32];
	uint32_t pfreq[19] = { 0 };
	uint8_t lens[288 + 32];
	static const uint8_t lextra[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
	static const uint8_t dextra[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };

	b->lf[256]++;                                   /* end of block */
	build_code(b->lf, 288, HD_LITLEN_MAXBITS, &lh);
	build_code(b->df, 32, HD_OFFSET_MAXBITS, &dh);
	unsigned hlit = 286, hdist = 30;
	while (hlit > 257 && lh.len[hlit - 1] == 0)
		hlit--;
	while (hdist > 1 && dh.len[hdist - 1] == 0)
		hdist--;
	memcpy(lens, lh.len, hlit);
	memcpy(lens + hlit, dh.len, hdist);
	unsigned ni = rle_lens(lens, hlit + hdist, items, pfreq);
	build_code(pfreq, 19, HD_PRECODE_MAXBITS, &ph);
	unsigned hclen = 19;
	while (hclen > 4 && ph.len[precode_perm[hclen - 1]] == 0)
		hclen--;
	/* exact costs: extra bits are common to both codes */
	uint64_t extra = 0, dyn = 3 + 5 + 5 + 4 + 3 * hclen, sta = 3;
	uint8_t sl[288], sd[32];
	static_lens(sl, sd);
	for (unsigned k = 0; k < ni; k++) {
		unsigned sym = items[k] & 31;
		dyn += ph.len[sym] + (sym == 16 ? 2 : sym == 17 ? 3 : sym == 18 ? 7 : 0);
	}
	for (unsigned s = 0; s < 286; s++) {
		dyn += (uint64_t)b->lf[s] * lh.len[s];
		sta += (uint64_t)b->lf[s] * sl[s];
		if (s >= 257)
			extra += (uint64_t)b->lf[s] * lextra[s - 257];
	}
	for (unsigned s = 0; s < 30; s++) {
		dyn += (uint64_t)b->df[s] * dh.len[s];
		sta += (uint64_t)b->df[s] * sd[s];
		extra += (uint64_t)b->df[s] * dextra[s];
	}
	int use_dynamic = dyn < sta;                    /* tie -> static (deflate_compress.c:1861-1867) */
	uint64_t blockbits = (use_dynamic ? dyn : sta) + extra;
	if (w->bitpos + blockbits > limit_bits)
		return 0;
	bw_put(w, final ? 1 : 0, 1);
	if (use_dynamic) {
		bw_put(w, 2, 2);
		bw_put(w, hlit - 257, 5);
		bw_put(w, hdist - 1, 5);
		bw_put(w, hclen - 4, 4);
		for (unsigned k = 0; k < hclen; k++)
			bw_put(w, ph.len[precode_perm[k]], 3);
		for (unsigned k = 0; k < ni; k++) {
			unsigned sym = items[k] & 31, ev = items[k] >> 8;
			bw_put(w, ph.code[sym], ph.len[sym]);
			if (sym == 16) bw_put(w, ev, 2);
			else if (sym == 17) bw_put(w, ev, 3);
			else if (sym == 18) bw_put(w, ev, 7);
		}
	} else {
		bw_put(w, 1, 2);
		static_lens(lh.len, dh.len);
		for (unsigned s = 0; s < 288; s++) {
			uint32_t c; unsigned nn;
			static_litlen(s, &c, &nn);
			lh.code[s] = (uint16_t)c;
		}
		for (unsigned s = 0; s < 32; s++)
			dh.code[s] = (uint16_t)bitrev(s, 5);
	}
	for (unsigned t = 0; t < b->ntok; t++) {
		uint32_t tk = b->tok[t];
		if (!(tk & HD_TOKEN_MATCH)) {
			bw_put(w, lh.code[tk & 0xff], lh.len[tk & 0xff]);
			continue;
		}
		unsigned sym, eb, ev;
		len_slot(((tk >> 16) & 0xff) + 3, &sym, &eb, &ev);
		bw_put(w, lh.code[sym], lh.len[sym]);
		bw_put(w, ev, eb);
		off_slot((tk & 0xffff) + 1, &sym, &eb, &ev);
		bw_put(w, dh.code[sym], dh.len[sym]);
		bw_put(w, ev, eb);
	}
	bw_put(w, lh.code[256], lh.len[256]);
	memset(b->lf, 0, sizeof(b->lf));
	memset(b->df, 0, sizeof(b->df));
	b->ntok = 0;
	return 1;
}

/* HD_FRAME_RAW_FLUSH: what zlibutil_buffer_full_flush (applet/7dictzip.c:93-126)
 * leaves behind a stream whose BFINAL bits it cleared: the 3 header bits of an
 * empty stored block in the unused bits of the last byte (one more zero byte
 * when fewer than 3 are unused, :117-119), then 00 00 ff ff (:120-123) */
static void put_flush_suffix(bw_t *w)
{
	bw_put(w, 0, 3);
	w->bitpos = (w->bitpos + 7) & ~(uint64_t)7;
	bw_put(w, 0xffff0000u, 32);
}

static int write_stored(uint8_t *dest, size_t *destLen, const uint8_t *in, size_t n, int flush)
{
	/* one BFINAL-terminated run of stored blocks; an empty input still gets its
	 * one empty stored block here (unlike store_deflate).  Flush form: no block
	 * is final, an empty stored block follows. */
	if (HD_STORED_SIZE(n) + (flush ? 5u : 0u) > *destLen)
		return 1;           /* !Z_OK, as libdeflate_deflate lib/zlibutil.c:189 */
	size_t o = 0, left = n;
	do {
		size_t blk = left < 65535 ? left : 65535;
		dest[o] = (left - blk || flush) ? 0 : 1;
		dest[o + 1] = blk & 0xff; dest[o + 2] = blk >> 8;
		dest[o + 3] = ~blk & 0xff; dest[o + 4] = (~blk >> 8) & 0xff;
		memcpy(dest + o + 5, in + (n - left), blk);
		o += 5 + blk;
		left -= blk;
	} while (left);
	if (flush) {
		static const uint8_t sfx[5] = { 0, 0, 0, 0xff, 0xff };
		memcpy(dest + o, sfx, 5);
		o += 5;
	}
	*destLen = o;
	return 0;
}

/* part: 0, or -- a latency-mode segment (HD_LAT_PARTS) -- the bytes each of its PARSE parts covers: every part is parsed on
 * its own (fresh table, fresh window, no match across the border: one wavefront each on the device), the tokens of all parts
 * then make ONE DEFLATE block with one code (the emit wavefront's) */
static int deflate_dynamic(uint8_t *dest, size_t *destLen, const uint8_t *in, size_t n,
			   unsigned win_bits, unsigned hash_bits, unsigned intra, int flush, unsigned part, size_t segprime)
{
	size_t cap = *destLen >= (flush ? 5u : 0u) ? *destLen - (flush ? 5u : 0u) : 0;
	size_t stored = HD_STORED_SIZE(n);
	size_t limit = cap < stored - 1 ? cap : stored - 1;
	uint8_t *tmp = calloc(1, limit + 64 + 8);
	mf_t mf = { calloc((size_t)1 << hash_bits, 2), hash_bits, 1u << win_bits, 0, intra };
	dynblk_t b;
	bw_t w = { tmp, 0 };
	step_t st;
	int alive = 1;

	memset(&b, 0, sizeof(b));
	b.tok = malloc((HD_DYN_BLOCK_TOKENS + 64) * 4);
	if (part)
		b.tok = realloc(b.tok, (n + HD_DYN_BLOCK_TOKENS + 64) * 4);     /* (never closed early: one block) */
	for (size_t ps = 0; ps < n && alive; ps += part ? part : n) {
		/* a part behind the first starts HD_LAT_PRIME_BYTES early; the tokens of those steps are dropped (HD_LAT_PRIME) */
		const size_t plen = part && n - ps > part ? part : n - ps;
		const size_t prime = !part ? 0 : ps ? HD_LAT_PRIME(ps, plen) : HD_LAT_PRIME(segprime, plen);   /* (the segment's own priming: its first part's) */
		const uint8_t *pin = in + ps - prime;
		const size_t pn = plen + prime;
		unsigned carry = 0;
		if (ps) {
			memset(mf.table, 0, ((size_t)1 << hash_bits) * 2);
			mf.filled = 0;
		}
		for (size_t S = 0; S < pn && alive; S += HD_WAVE) {
			if (S == prime)
				carry = 0;
			parse_step(&mf, pin, pn, S, carry, &st);
			carry = st.carry_out;
			if (S < prime)
				continue;
			for (unsigned l = 0; l < st.lanes; l++) {
				if (st.is_match[l]) {
					unsigned sym, eb, ev;
					b.tok[b.ntok++] = HD_TOKEN_MATCH | ((uint32_t)(st.len[l] - 3) << 16) | (st.dist[l] - 1);
					len_slot(st.len[l], &sym, &eb, &ev);
					b.lf[sym]++;
					off_slot(st.dist[l], &sym, &eb, &ev);
					b.df[sym]++;
				} else if (st.is_lit[l]) {
					b.tok[b.ntok++] = pin[S + l];
					b.lf[pin[S + l]]++;
				}
			}
			if (!part && b.ntok >= HD_DYN_BLOCK_TOKENS && S + HD_WAVE < n)
				alive = flush_dyn_block(&w, &b, 0, 8 * (uint64_t)limit);
		}
	}
	if (alive)
		alive = flush_dyn_block(&w, &b, !flush, 8 * (uint64_t)limit);
	int ret;
	if (alive) {
		if (flush)
			put_flush_suffix(&w);
		*destLen = (size_t)((w.bitpos + 7) >> 3);
		memcpy(dest, tmp, *destLen);
		ret = 0;
	} else {
		ret = write_stored(dest, destLen, in, n, flush);
	}
	free(b.tok);
	free(mf.table);
	free(tmp);
	return ret;
}


/* ---- levels >= HD_WG_LEVEL, throughput form: the workgroup parse (include/hipdeflate_params.h "WORKGROUP LEVELS") ------
 * The serial statement of hd_deflate_wg.hpp.  Roles: hc_matchfinder_longest_match (lib/libdeflate/hc_matchfinder.h:183-338)
 * -> the bucket's four positions + the run candidate, verified over HD_WG_VCAP bytes; the lazy rule and the token choice of
 * deflate_compress_lazy_generic (deflate_compress.c:2606-2809) -> steps 5..7 below; should_end_block (:2141-2218) -> split. */
typedef struct {
	uint32_t obs[3], nobs[3], n, nn;        /* literal / match of < 9 bytes / longer match: merged, and since the last check */
} wg_split_t;

static int wg_split_check(wg_split_t *s, uint32_t block_len)
{
	if (s->n > 0) {
		/* sum of absolute differences of the class probabilities, times n * nn (do_end_block_check,
		 * deflate_compress.c:2143-2196), in 32-bit arithmetic: counts are below 2^17, nn below 2^10 */
		uint32_t total = 0;
		for (int i = 0; i < 3; i++) {
			const uint32_t e = s->obs[i] * s->nn, a = s->nobs[i] * s->n;
			total += a > e ? a - e : e - a;
		}
		const uint32_t items = s->n + s->nn;
		uint32_t cutoff = s->nn * 200u / 512u * s->n;
		if (block_len < 10000 && items < 8192)
			cutoff += (cutoff >> 13) * (8192u - items);
		if (total + (block_len / 4096u) * s->n >= cutoff)
			return 1;
	}
	for (int i = 0; i < 3; i++) {
		s->obs[i] += s->nobs[i];
		s->nobs[i] = 0;
	}
	s->n += s->nn;
	s->nn = 0;
	return 0;
}

static int deflate_wg(uint8_t *dest, size_t *destLen, const uint8_t *in, size_t n, int flush, int level)
{
	const unsigned WAYS = HD_WG_WAYS(level), BUCKETS = HD_WG_BUCKETS(level);
	const int lazy = HD_WG_LAZY(level);
	const size_t cap = *destLen >= (flush ? 5u : 0u) ? *destLen - (flush ? 5u : 0u) : 0;
	const size_t stored = HD_STORED_SIZE(n);
	const size_t limit = cap < stored - 1 ? cap : stored - 1;
	/* (round 5: a block longer than its room is coded like any other and goes through when its stream fits -- the contract of
	 * libdeflate_deflate, lib/zlibutil.c:179-192, which applet/7png.c:112 leans on with 1.5 x the OLD compressed size as
	 * room; round 4 refused such a block because the kernel's records were sized by the slot) */
	uint8_t *tmp = calloc(1, limit + 64 + 8);
	uint16_t *bk = calloc((size_t)BUCKETS * WAYS, 2);
	dynblk_t b;
	bw_t w = { tmp, 0 };
	wg_split_t sp;
	int alive = 1;
	uint32_t clen[HD_WAVE], flen[HD_WAVE], dist[HD_WAVE];
	uint16_t pre[HD_WAVE][HD_WG_MAX_WAYS];
	size_t E = 0, block_begin = 0;

	memset(&b, 0, sizeof(b));
	memset(&sp, 0, sizeof(sp));
	b.tok = malloc((n + 64) * 4);
	for (size_t S = 0; S < n && alive; S += HD_WAVE) {
		const unsigned lanes = n - S < HD_WAVE ? (unsigned)(n - S) : HD_WAVE;
		/* 1. every lane with six bytes left reads its bucket as the steps before left it */
		for (unsigned l = 0; l < lanes; l++) {
			const size_t p = S + l;
			memset(pre[l], 0, sizeof(pre[l]));
			if (p + HD_LAZY_KEY_BYTES <= n)
				memcpy(pre[l], bk + (size_t)HD_HASH_SLOT6(load32(in + p), in[p + 4] | (in[p + 5] << 8), BUCKETS) * WAYS,
				       2 * WAYS);
		}
		/* 2. ... and stores { itself, the three newest before the step }: in lane order, so the highest lane's store stays */
		for (unsigned l = 0; l < lanes; l++) {
			const size_t p = S + l;
			if (p + HD_LAZY_KEY_BYTES > n)
				continue;
			uint16_t *e = bk + (size_t)HD_HASH_SLOT6(load32(in + p), in[p + 4] | (in[p + 5] << 8), BUCKETS) * WAYS;
			e[0] = (uint16_t)p;             /* the position mod 2^16 = its offset in the 64 KiB ring; nothing marks an entry
			                                 * empty: whatever it holds names a position, and the bytes there decide */
			for (unsigned k = 1; k < WAYS; k++)
				e[k] = pre[l][k - 1];
		}
		/* 3. verify: the byte before (runs; only inside the step), then the bucket, newest first; longest wins, nearest on a tie */
		for (unsigned l = 0; l < lanes; l++) {
			const size_t p = S + l;
			/* no match crosses a multiple of HD_WG_CUT: the parse of one HD_WG_CUT-byte piece never depends on another's */
			const unsigned to_cut = HD_WG_CUT - (unsigned)(p % HD_WG_CUT);
			unsigned room = n - p < HD_WG_VCAP ? (unsigned)(n - p) : HD_WG_VCAP;
			if (to_cut < room)
				room = to_cut;
			unsigned best = 0;
			uint32_t bd = 0;
			clen[l] = flen[l] = dist[l] = 0;
			if (p + HD_LAZY_KEY_BYTES > n)
				continue;
			for (unsigned k = 0; k <= WAYS; k++) {
				uint32_t back;
				if (k == 0) {
					if (l == 0)
						continue;
					back = 1;
				} else {
					const uint32_t e = pre[l][k - 1];
					back = (uint32_t)(p - e) & 0xffffu;
					if (back == 0 || back > HD_WG_WINDOW || back > p)
						continue;
				}
				unsigned m = 0;
				while (m < room && in[p + m] == in[p - back + m])
					m++;
				if (m > best) {
					best = m;
					bd = back;
				}
			}
			if (best < HD_WG_MIN_LEN)
				continue;
			clen[l] = best;
			dist[l] = bd;
			/* 4. a match of the whole verified span is extended to its full length */
			unsigned len = best;
			if (best == HD_WG_VCAP) {
				unsigned maxlen = n - p < HD_MAX_MATCH ? (unsigned)(n - p) : HD_MAX_MATCH;
				if (to_cut < maxlen)
					maxlen = to_cut;
				while (len < maxlen && in[p + len] == in[p + len - bd])
					len++;
			}
			flen[l] = len;
		}
		/* 5. the lazy rule on the right-hand neighbour, 6. the path from E, 7. tokens */
		for (unsigned l = 0; l < lanes; l++) {
			const size_t p = S + l;
			if (p < E)
				continue;
			int take = clen[l] != 0;
			if (take && lazy && l + 1 < lanes && clen[l + 1] >= clen[l] && clen[l + 1] &&
			    4 * ((int)clen[l + 1] - (int)clen[l]) + ((int)ilog2(dist[l]) - (int)ilog2(dist[l + 1])) > 2)
				take = 0;
			if (take) {
				unsigned sym, eb, ev;
				b.tok[b.ntok++] = HD_TOKEN_MATCH | ((uint32_t)(flen[l] - 3) << 16) | (dist[l] - 1);
				len_slot(flen[l], &sym, &eb, &ev);
				b.lf[sym]++;
				off_slot(dist[l], &sym, &eb, &ev);
				b.df[sym]++;
				E = p + flen[l];
				sp.nobs[flen[l] >= 9 ? 2 : 1]++;
			} else {
				b.tok[b.ntok++] = in[p];
				b.lf[in[p]]++;
				E = p + 1;
				sp.nobs[0]++;
			}
			sp.nn++;
		}
		/* a DEFLATE block may end behind any piece of HD_WG_CUT bytes but the last */
		const size_t here = S + lanes;
		if (here < n && here % HD_WG_CUT == 0) {
			int end = b.ntok >= HD_DYN_BLOCK_TOKENS;
			if (!end && sp.nn >= HD_WG_SPLIT_OBS && here - block_begin >= HD_WG_SPLIT_MIN && n - here >= HD_WG_SPLIT_MIN)
				end = wg_split_check(&sp, (uint32_t)(here - block_begin));
			if (end) {
				alive = flush_dyn_block(&w, &b, 0, 8 * (uint64_t)limit);
				memset(&sp, 0, sizeof(sp));
				block_begin = here;
			}
		}
	}
	if (alive)
		alive = flush_dyn_block(&w, &b, !flush, 8 * (uint64_t)limit);
	int ret;
	if (alive) {
		if (flush)
			put_flush_suffix(&w);
		*destLen = (size_t)((w.bitpos + 7) >> 3);
		memcpy(dest, tmp, *destLen);
		ret = 0;
	} else {
		ret = write_stored(dest, destLen, in, n, flush);
	}
	free(b.tok);
	free(bk);
	free(tmp);
	return ret;
}

/* test hook: the code lengths build_code() gives a frequency vector (tests/test_oracle_golden.py checks
 * Kraft equality and the length limit on adversarial distributions) */
void hdo_build_lengths(const uint32_t *freq, unsigned nsyms, unsigned maxbits, uint8_t *lens_out)
{
	huff_t h;
	build_code(freq, nsyms, maxbits, &h);
	memcpy(lens_out, h.len, nsyms);
}

static int twin(uint8_t *dest, size_t *destLen, const uint8_t *source, size_t sourceLen, int level, int flush, unsigned lat,
		unsigned part, size_t prime);

/* levels >= 1, blocks longer than HD_SEG_LIMIT (include/hipdeflate_params.h) -- or, in latency mode, longer than
 * `seg` = HD_LAT_SEG_BYTES(level): independent `seg`-byte segments in flush form one behind the other, then the empty
 * final block unless the member itself is a flush form.  The room has to cover the worst case of every segment,
 * whatever the data turns out to need. */
static int twin_segmented(uint8_t *dest, size_t *destLen, const uint8_t *source, size_t sourceLen, int level, int flush,
			  unsigned seg)
{
	if (*destLen < HD_SEGN_WORST((uint64_t)sourceLen, seg, flush))
		return 1;
	size_t o = 0;
	for (size_t s = 0; s < sourceLen; s += seg) {
		size_t n = sourceLen - s < seg ? sourceLen - s : seg;
		size_t room = *destLen - o;
		/* latency segments behind the first are primed with the end of their predecessor (HD_LAT_PRIME) */
		int r = twin(dest + o, &room, source + s, n, level, 1, 0, HD_LAT_PARTS(level, seg) ? HD_LAT_PART_BYTES : 0,
			     (HD_LAT_SEG_PRIME && s && seg != HD_SEG_BYTES) ? HD_LAT_PRIME(HD_LAT_PRIME_BYTES, n) : 0);
		if (r)
			return r;
		o += room;
	}
	if (!flush) {
		dest[o++] = 0x03;
		dest[o++] = 0x00;
	}
	*destLen = o;
	return 0;
}

/* lat: latency mode (HD_FRAME_LATENCY) */
static int twin(uint8_t *dest, size_t *destLen, const uint8_t *source, size_t sourceLen, int level, int flush, unsigned lat,
		unsigned part, size_t prime)
{
	/* (the per-block codecs and the hook take the latency form only when the room covers its worst case, and the
	 * ordinary form otherwise -- as libdeflate_deflate they succeed whenever the stored form fits) */
	/* the workgroup levels take a block of any length as one stream, in EITHER mode: one codec per level (round 5; the
	 * reference's deflate_compress.c:3951-3955 -- what bgzf_compress.c:163-169 and lib/zlibutil.c:179-192 call is what
	 * applet/7bgzf.c's loop calls).  On the device latency mode only changes who writes the member (hd_emit_wg.hpp) */
	if (level >= HD_WG_LEVEL)
		return deflate_wg(dest, destLen, source, sourceLen, flush, level);
	if (level >= 1 && lat && sourceLen > HD_LAT_SEG_BYTES(level) &&
	    *destLen >= HD_SEGN_WORST((uint64_t)sourceLen, HD_LAT_SEG_BYTES(level), flush))
		return twin_segmented(dest, destLen, source, sourceLen, level, flush, HD_LAT_SEG_BYTES(level));
	if (level >= 1 && sourceLen > HD_SEG_LIMIT)
		return twin_segmented(dest, destLen, source, sourceLen, level, flush, HD_SEG_BYTES);
	if (level <= 0)
		return write_stored(dest, destLen, source, sourceLen, flush);   /* level 0 = the stored branch */
	if (level == 1)
		return deflate_static(dest, destLen, source, sourceLen, HD_L1_WIN_BITS, HD_L1_HASH_BITS, flush, prime);
	if (level == 2)
		return deflate_dynamic(dest, destLen, source, sourceLen, HD_L2_WIN_BITS, HD_L2_HASH_BITS, HD_INTRA_DIST, flush, part, prime);
	return write_stored(dest, destLen, source, sourceLen, flush);   /* (not reached) */
}

int hdo_deflate_twin(uint8_t *dest, size_t *destLen, const uint8_t *source,
		     size_t sourceLen, int level)
{
	return twin(dest, destLen, source, sourceLen, level, 0, 0, 0, 0);
}

/* the same encoder in HD_FRAME_RAW_FLUSH form (include/hipdeflate.h) */
int hdo_deflate_twin_flush(uint8_t *dest, size_t *destLen, const uint8_t *source,
			   size_t sourceLen, int level)
{
	return twin(dest, destLen, source, sourceLen, level, 1, 0, 0, 0);
}

/* ... in latency mode (HD_FRAME_LATENCY): what bgzf_compress, hip_deflate and hip_deflate_flush produce */
int hdo_deflate_twin_lat(uint8_t *dest, size_t *destLen, const uint8_t *source, size_t sourceLen, int level)
{
	return twin(dest, destLen, source, sourceLen, level, 0, 1, 0, 0);
}

int hdo_deflate_twin_lat_flush(uint8_t *dest, size_t *destLen, const uint8_t *source, size_t sourceLen, int level)
{
	return twin(dest, destLen, source, sourceLen, level, 1, 1, 0, 0);
}
