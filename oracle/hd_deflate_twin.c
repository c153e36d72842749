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
	uint16_t *table;      /* 1 << hash_bits entries: (position + 1) mod 2^16, 0 = empty */
	unsigned hash_bits;
	unsigned win;         /* ring size in bytes */
	size_t filled;        /* bytes the GPU ring has been filled up to */
} mf_t;

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

static void parse_step(mf_t *mf, const uint8_t *in, size_t n, size_t S, unsigned carry, step_t *st)
{
	uint32_t cand[HD_WAVE];
	uint8_t ok[HD_WAVE];
	unsigned lanes = n - S < HD_WAVE ? (unsigned)(n - S) : HD_WAVE;

	/* the ring is refilled a 1 KiB piece at a time until it holds
	 * HD_LOOKAHEAD bytes past S (or the whole input) */
	while (mf->filled < n && mf->filled < S + HD_LOOKAHEAD)
		mf->filled += HD_PIECE;
	size_t lo = mf->filled > mf->win ? mf->filled - mf->win : 0;

	memset(st, 0, sizeof(*st));
	for (unsigned l = 0; l < lanes; l++) {          /* 1. look up */
		size_t p = S + l;
		ok[l] = 0;
		if (p + HD_MIN_MATCH > n)
			continue;
		uint32_t v = load32(in + p);
		uint32_t e = mf->table[(v * HD_HASH_MUL) >> (32 - mf->hash_bits)];
		/* the latest p' < p with p' + 1 == e (mod 2^16); for inputs <= 64 KiB
		 * that is simply e */
		uint32_t back = (uint32_t)(p + 1 - e) & 0xffffu;   /* 0 = exactly 2^16 back: stale */
		cand[l] = (e && back) ? (uint32_t)(p + 1 - back) : 0;
	}
	for (unsigned l = 0; l < lanes; l++) {          /* 2. publish */
		size_t p = S + l;
		if (p + HD_MIN_MATCH > n)
			continue;
		/* the kernel's lanes race for the slot and re-write until the largest
		 * position of the step holds it: within a step the last lane wins */
		mf->table[(load32(in + p) * HD_HASH_MUL) >> (32 - mf->hash_bits)] = (uint16_t)(p + 1);
	}
	for (unsigned l = 0; l < lanes; l++) {          /* 3. verify */
		size_t p = S + l;
		if (p + HD_MIN_MATCH > n || cand[l] == 0)
			continue;
		size_t c = cand[l] - 1;
		if (c < lo)
			continue;
		if (load32(in + c) != load32(in + p))
			continue;
		ok[l] = 1;
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

/* ---- level 1: greedy + static Huffman, streaming ------------------------ */
static int deflate_static(uint8_t *dest, size_t *destLen, const uint8_t *in, size_t n,
			  unsigned win_bits, unsigned hash_bits)
{
	size_t cap = *destLen;
	size_t stored = HD_STORED_SIZE(n);
	/* the static stream is kept only if it ends up strictly smaller than the
	 * stored form and fits; tie -> stored, as deflate_flush_block's
	 * "stored > static > dynamic" preference (deflate_compress.c:1820-1867) */
	size_t limit = cap < stored - 1 ? cap : stored - 1;
	uint8_t *tmp = calloc(1, limit + HD_STEP_MAX_BITS / 8 + 16);
	mf_t mf = { calloc((size_t)1 << hash_bits, 2), hash_bits, 1u << win_bits, 0 };
	bw_t w = { tmp, 0 };
	int use_static = 1;
	step_t st;
	uint8_t scratch[HD_STEP_MAX_BITS / 8 + 8];

	bw_put(&w, 1, 1);       /* BFINAL */
	bw_put(&w, 1, 2);       /* BTYPE = 01 */
	unsigned carry = 0;
	for (size_t S = 0; S < n && use_static; S += HD_WAVE) {
		parse_step(&mf, in, n, S, carry, &st);
		carry = st.carry_out;
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
		*destLen = (size_t)((w.bitpos + 7) >> 3);
		memcpy(dest, tmp, *destLen);
	} else if (stored <= cap) {
		/* one BFINAL-terminated run of stored blocks; an empty input still
		 * gets its one empty stored block here (unlike store_deflate) */
		size_t o = 0, left = n;
		do {
			size_t blk = left < 65535 ? left : 65535;
			dest[o] = left - blk ? 0 : 1;
			dest[o + 1] = blk & 0xff; dest[o + 2] = blk >> 8;
			dest[o + 3] = ~blk & 0xff; dest[o + 4] = (~blk >> 8) & 0xff;
			memcpy(dest + o + 5, in + (n - left), blk);
			o += 5 + blk;
			left -= blk;
		} while (left);
		*destLen = o;
	} else {
		ret = 1;            /* !Z_OK, as libdeflate_deflate lib/zlibutil.c:189 */
	}
	free(mf.table);
	free(tmp);
	return ret;
}

int hdo_deflate_twin(uint8_t *dest, size_t *destLen, const uint8_t *source,
		     size_t sourceLen, int level)
{
	if (level <= 0) {
		size_t stored = HD_STORED_SIZE(sourceLen);
		if (stored > *destLen)
			return 1;
		size_t cap0 = 0; /* force the stored branch */
		(void)cap0;
		/* level 0 = the stored branch of the level-1 encoder */
		size_t o = 0, left = sourceLen;
		do {
			size_t blk = left < 65535 ? left : 65535;
			dest[o] = left - blk ? 0 : 1;
			dest[o + 1] = blk & 0xff; dest[o + 2] = blk >> 8;
			dest[o + 3] = ~blk & 0xff; dest[o + 4] = (~blk >> 8) & 0xff;
			memcpy(dest + o + 5, source + (sourceLen - left), blk);
			o += 5 + blk;
			left -= blk;
		} while (left);
		*destLen = o;
		return 0;
	}
	return deflate_static(dest, destLen, source, sourceLen,
			      HD_L1_WIN_BITS, HD_L1_HASH_BITS);
}
