/* mf_explore.c -- research tool (CPU, not product, not oracle): what a matchfinder that a WORKGROUP can hold in LDS
 * buys in compressed bytes, before anything is written for the GPU.  A parametric restatement of the twin's step parse
 * (oracle/hd_deflate_twin.c parse_step) -- G positions per step that cannot see one another, B buckets of W positions
 * (newest first, one new entry per bucket and step: the highest lane's store stays), a K-byte hash key, candidates
 * verified over VCAP bytes, greedy / neighbour-lazy / libdeflate's lazy rule, the minimum-length table of
 * deflate_compress.c:2296-2379, block splitting by the observation test of :2141-2218 -- with the twin's own Huffman
 * construction and block writer behind it (the file is #included), so the byte counts are exact DEFLATE sizes.
 *
 *   mf_explore FILE BLOCK [key=value ...]        prints the total compressed bytes of FILE cut into BLOCK-byte blocks
 *   keys: win (bits) buckets ways key G intra vcap minlen lazy(0/1/2) adapt split seg prime toks h3
 * tools/mf_grid.py runs the grids quoted in DESIGN.md against tests/golden/ratio_ref.json. */
#define main twin_unused_main
#include "../oracle/hd_deflate_twin.c"
#undef main
#include <stdio.h>

typedef struct {
	unsigned win_bits, buckets, ways, key, G, intra, vcap, minlen, lazy, adapt, split, seg, prime, toks, h3, far5, ins_all, k2, b2, lazyd, cut, splitg, rr;
} cfg_t;

static uint32_t hash_key(const uint8_t *p, unsigned key, unsigned buckets)
{
	uint64_t v = 0;
	for (unsigned i = 0; i < key; i++)
		v |= (uint64_t)p[i] << (8 * i);
	v *= 0x9E3779B185EBCA87ull;
	return (uint32_t)((v >> 40) * (uint64_t)buckets >> 24);
}

static unsigned bsr32(uint32_t v) { return 31 - (unsigned)__builtin_clz(v); }

static const uint8_t min_lens_tab[] = { 9, 9, 9, 9, 9, 9, 8, 8, 7, 7, 6, 6, 6, 6, 6, 6, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5,
					5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
					4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4 };
static unsigned choose_min_len(unsigned used) { return used < sizeof(min_lens_tab) ? min_lens_tab[used] : 3; }

typedef struct {
	uint32_t obs[10], nobs[10], n, nn;
} split_t;

static int split_check(split_t *s, uint32_t block_len)
{
	if (s->n > 0) {
		uint64_t total = 0;
		for (int i = 0; i < 10; i++) {
			uint64_t e = (uint64_t)s->obs[i] * s->nn, a = (uint64_t)s->nobs[i] * s->n;
			total += a > e ? a - e : e - a;
		}
		uint64_t items = s->n + s->nn, cutoff = (uint64_t)s->nn * 200 / 512 * s->n;
		if (block_len < 10000 && items < 8192)
			cutoff += cutoff * (8192 - items) / 8192;
		if (total + (uint64_t)(block_len / 4096) * s->n >= cutoff)
			return 1;
	}
	for (int i = 0; i < 10; i++) {
		s->obs[i] += s->nobs[i];
		s->nobs[i] = 0;
	}
	s->n += s->nn;
	s->nn = 0;
	return 0;
}

/* one segment: tokens into the DEFLATE blocks of `w`; prime = bytes before `in` that only fill the tables */
static int code_segment(bw_t *w, const uint8_t *in, size_t n, size_t prime, const cfg_t *c, int final, uint64_t limit_bits)
{
	const unsigned W = c->ways, B = c->buckets, G = c->G;
	uint32_t *bk = calloc((size_t)B * W, 4);               /* position + 1, 0 = empty */
	uint32_t *h3 = c->h3 ? calloc(1u << 15, 4) : NULL;
	uint32_t *t2 = c->k2 ? calloc(c->b2, 4) : NULL;
	dynblk_t b;
	memset(&b, 0, sizeof(b));
	b.tok = malloc((n + 64) * 4);
	split_t sp;
	memset(&sp, 0, sizeof(sp));
	const uint8_t *base = in - prime;
	const size_t tot = n + prime;
	uint32_t *cand_len = malloc(G * 4), *cand_dist = malloc(G * 4);
	size_t E = 0;                                          /* first position not yet covered */
	size_t block_begin = prime;
	unsigned minlen = c->minlen;
	int alive = 1;
	if (c->adapt) {
		uint8_t used[256] = { 0 };
		unsigned nu = 0;
		for (size_t i = 0; i < (n < 4096 ? n : 4096); i++)
			used[in[i]] = 1;
		for (int i = 0; i < 256; i++)
			nu += used[i];
		minlen = n < 512 ? 3 : choose_min_len(nu);
		if (minlen < c->minlen)
			minlen = c->minlen;
	}
	size_t next_recalc = prime + 10000;
	for (size_t S = 0; S < tot && alive; S += G) {
		const unsigned lanes = tot - S < G ? (unsigned)(tot - S) : G;
		const size_t lo = S + lanes > (1u << c->win_bits) ? S + lanes - (1u << c->win_bits) : 0;   /* window: as the ring holds it */
		/* 1. look up + verify */
		for (unsigned l = 0; l < lanes; l++) {
			const size_t p = S + l;
			cand_len[l] = 0;
			cand_dist[l] = 0;
			if (p + c->key > tot)
				continue;
			unsigned room = tot - p < c->vcap ? (unsigned)(tot - p) : c->vcap;
			if (c->cut && c->cut - (unsigned)(p % c->cut) < room)   /* no match crosses a cut boundary */
				room = c->cut - (unsigned)(p % c->cut);
			unsigned best = 0;
			uint32_t bd = 0;
			const uint32_t *e = bk + (size_t)hash_key(base + p, c->key, B) * W;
			for (unsigned k = 0; k <= W + c->intra + 1; k++) {
				size_t q;
				if (k < c->intra) {                     /* same-step neighbours first (nearest) */
					if (k + 1 > l)
						continue;
					q = p - (k + 1);
				} else if (k < c->intra + W) {
					if (!e[k - c->intra])
						continue;
					q = e[k - c->intra] - 1;
				} else if (k == c->intra + W && h3) {
					const uint32_t v = h3[((base[p] | base[p + 1] << 8 | base[p + 2] << 16) * 0x9E3779B1u) >> 17];
					if (!v)
						continue;
					q = v - 1;
				} else if (k == c->intra + W + 1 && t2) {
					const uint32_t v = t2[hash_key(base + p, c->k2, c->b2)];
					if (!v)
						continue;
					q = v - 1;
				} else {
					continue;
				}
				if (q < lo || q >= p)
					continue;
				unsigned m = 0;
				while (m < room && base[p + m] == base[q + m])
					m++;
				if (m > best || (c->rr && m == best && m && (uint32_t)(p - q) < bd)) {   /* newest first: an older one only when strictly longer (rr: the ways are in no order, the nearer wins a tie) */
					best = m;
					bd = (uint32_t)(p - q);
				}
			}
			if (best >= minlen && !(best == 3 && bd > 8192) && !(c->far5 && best <= c->far5 && bd > 16384)) {
				cand_len[l] = best;
				cand_dist[l] = bd;
			}
		}
		/* 2. publish: per bucket the last position of the step (ins_all: every position, as a serial matchfinder) */
		for (unsigned l = 0; l < lanes; l++) {
			const size_t p = S + l;
			if (p + c->key > tot)
				continue;
			const uint32_t hb = hash_key(base + p, c->key, B);
			int last = 1;
			if (!c->ins_all)
				for (unsigned m = l + 1; m < lanes && last; m++)
					if (S + m + c->key <= tot && hash_key(base + S + m, c->key, B) == hb)
						last = 0;
			if (last) {
				uint32_t *e = bk + (size_t)hb * W;
				if (c->rr) {                            /* rr: the way is the step's number mod W -- a store that needs no read */
					e[(S / G) % W] = (uint32_t)(p + 1);
				} else {
					memmove(e + 1, e, (W - 1) * 4);
					e[0] = (uint32_t)(p + 1);
				}
			}
			if (h3)
				h3[((base[p] | base[p + 1] << 8 | base[p + 2] << 16) * 0x9E3779B1u) >> 17] = (uint32_t)(p + 1);
			if (t2)
				t2[hash_key(base + p, c->k2, c->b2)] = (uint32_t)(p + 1);
		}
		/* 3. resolve */
		for (unsigned l = 0; l < lanes; l++) {
			const size_t p = S + l;
			if (p < E)
				continue;
			int take = cand_len[l] != 0;
			if (take && c->lazy == 1 && l + 1 < lanes && cand_len[l + 1] > cand_len[l])
				take = 0;
			if (take && c->lazy == 2 && l + 1 < lanes && cand_len[l + 1] >= cand_len[l] && cand_len[l + 1] &&
			    4 * ((int)cand_len[l + 1] - (int)cand_len[l]) + ((int)bsr32(cand_dist[l]) - (int)bsr32(cand_dist[l + 1])) > 2)
				take = 0;
			if (p < prime) {                                /* priming steps: tables only */
				if (take) {
					unsigned len = cand_len[l];
					const unsigned maxlen = tot - p < 258 ? (unsigned)(tot - p) : 258;
					while (len < maxlen && base[p + len] == base[p + len - cand_dist[l]])
						len++;
					E = p + len;
				} else {
					E = p + 1;
				}
				if (E > prime)
					E = prime;                      /* no match crosses into the segment */
				continue;
			}
			if (take) {
				unsigned len = cand_len[l], sym, eb, ev;
				unsigned maxlen = tot - p < 258 ? (unsigned)(tot - p) : 258;
				if (c->cut && c->cut - (unsigned)(p % c->cut) < maxlen)
					maxlen = c->cut - (unsigned)(p % c->cut);
				while (len < maxlen && base[p + len] == base[p + len - cand_dist[l]])
					len++;
				b.tok[b.ntok++] = HD_TOKEN_MATCH | ((uint32_t)(len - 3) << 16) | (cand_dist[l] - 1);
				len_slot(len, &sym, &eb, &ev);
				b.lf[sym]++;
				off_slot(cand_dist[l], &sym, &eb, &ev);
				b.df[sym]++;
				E = p + len;
				sp.nobs[8 + (len >= 9)]++;
				sp.nn++;
			} else {
				b.tok[b.ntok++] = base[p];
				b.lf[base[p]]++;
				E = p + 1;
				sp.nobs[c->split == 2 ? 0 : ((base[p] >> 5) & 6) | (base[p] & 1)]++;
				sp.nn++;
			}
		}
		if (S + lanes < prime)
			continue;
		const size_t here = S + lanes;                      /* the block can end at a step boundary */
		if (c->adapt && here >= next_recalc) {
			uint32_t lf = 0, nu = 0;
			for (int i = 0; i < 256; i++)
				lf += b.lf[i];
			for (int i = 0; i < 256; i++)
				nu += b.lf[i] > (lf >> 10);
			minlen = choose_min_len(nu);
			if (minlen < c->minlen)
				minlen = c->minlen;
			next_recalc += here - block_begin < tot - next_recalc ? here - block_begin : tot - next_recalc;
		}
		int end = 0;
		if (here < tot && !(c->splitg && here % c->splitg)) {      /* splitg: blocks end at multiples of it only */
			if (b.ntok >= c->toks)
				end = 1;
			else if (c->split && sp.nn >= 512 && here - block_begin >= 5000 && tot - here >= 5000)
				end = split_check(&sp, (uint32_t)(here - block_begin));
		}
		if (end) {
			/* tokens that reach past `here` stay with this block: E may be beyond */
			alive = flush_dyn_block(w, &b, 0, limit_bits);
			memset(&sp, 0, sizeof(sp));
			block_begin = here;
		}
	}
	if (alive)
		alive = flush_dyn_block(w, &b, final, limit_bits);
	free(b.tok);
	free(bk);
	free(h3);
	free(t2);
	free(cand_len);
	free(cand_dist);
	return alive;
}

static size_t code_block(const uint8_t *in, size_t n, const cfg_t *c)
{
	uint8_t *tmp = calloc(1, n + n / 2 + 4096);
	bw_t w = { tmp, 0 };
	const size_t seg = c->seg && n > c->seg ? c->seg : n;
	int alive = 1;
	for (size_t s = 0; s < n && alive; s += seg) {
		const size_t len = n - s < seg ? n - s : seg;
		const size_t prime = s < c->prime ? s : c->prime;
		alive = code_segment(&w, in + s, len, prime, c, seg == n, 8 * (uint64_t)(n + n / 2));
		if (seg != n)
			put_flush_suffix(&w);
	}
	size_t out = (size_t)((w.bitpos + 7) >> 3) + (seg != n ? 2 : 0);
	const size_t stored = HD_STORED_SIZE(n);
	if (!alive || out >= stored)
		out = stored;
	/* check: the stream must inflate to the input */
	if (alive && out < stored) {
		if (seg != n) {
			tmp[out - 2] = 3;
			tmp[out - 1] = 0;
		}
		uint8_t *back = malloc(n + 16);
		size_t bl = n;
		int st = hdo_inflate(back, &bl, tmp, out, NULL);
		if (st || bl != n || memcmp(back, in, n)) {
			fprintf(stderr, "mf_explore: round trip FAILED (status %d)\n", st);
			exit(2);
		}
		free(back);
	}
	free(tmp);
	return out;
}

int main(int argc, char **argv)
{
	if (argc < 3) {
		fprintf(stderr, "usage: %s FILE BLOCK [key=value ...]\n", argv[0]);
		return 2;
	}
	cfg_t c = { 15, 8192, 2, 6, 64, 1, 16, 5, 1, 0, 0, 0, 0, 32768, 0, 0, 0, 0, 4096, 0 };
	for (int i = 3; i < argc; i++) {
		char *eq = strchr(argv[i], '=');
		if (!eq)
			continue;
		*eq = 0;
		const unsigned v = (unsigned)atoi(eq + 1);
		const char *k = argv[i];
#define KEY(name, field) if (!strcmp(k, name)) c.field = v
		KEY("win", win_bits); KEY("buckets", buckets); KEY("ways", ways); KEY("key", key); KEY("G", G); KEY("intra", intra);
		KEY("vcap", vcap); KEY("minlen", minlen); KEY("lazy", lazy); KEY("adapt", adapt); KEY("split", split); KEY("seg", seg);
		KEY("prime", prime); KEY("toks", toks); KEY("h3", h3); KEY("far5", far5); KEY("ins_all", ins_all); KEY("k2", k2); KEY("b2", b2); KEY("lazyd", lazyd); KEY("cut", cut); KEY("splitg", splitg); KEY("rr", rr);
	}
	FILE *f = fopen(argv[1], "rb");
	if (!f)
		return 2;
	fseek(f, 0, SEEK_END);
	const size_t total = (size_t)ftell(f);
	fseek(f, 0, SEEK_SET);
	uint8_t *data = malloc(total);
	if (fread(data, 1, total, f) != total)
		return 2;
	fclose(f);
	const size_t block = (size_t)atol(argv[2]);
	size_t out = 0;
	for (size_t o = 0; o < total; o += block)
		out += code_block(data + o, total - o < block ? total - o : block, &c);
	printf("%zu\n", out);
	return 0;
}
