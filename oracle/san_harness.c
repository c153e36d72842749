/* san_harness.c -- runs the oracle (twin encoder at every level, inflate, checksums,
 * framing) over seeded inputs under -fsanitize=address,undefined.  TEST ONLY. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "hd_oracle.h"

static uint32_t rng = 12345;
static uint32_t rnd(void) { rng = rng * 1664525u + 1013904223u; return rng >> 8; }

int main(void)
{
	static const size_t sizes[] = { 0, 1, 3, 4, 5, 63, 64, 65, 100, 1000, 4095, 4096, 65280, 65536, 200000 };
	int fails = 0;
	for (unsigned si = 0; si < sizeof(sizes) / sizeof(sizes[0]); si++) {
		size_t n = sizes[si];
		for (int kind = 0; kind < 4; kind++) {
			uint8_t *in = malloc(n + 1), *z = malloc(n + n / 2 + 1024), *back = malloc(n + 1);
			for (size_t i = 0; i < n; i++)
				in[i] = kind == 0 ? (uint8_t)rnd() : kind == 1 ? "ACGT"[rnd() & 3] : kind == 2 ? 0
					: (uint8_t)("the quick brown fox "[i % 20] + ((rnd() & 63) == 0));
			for (int level = 0; level <= 9; level += (level < 2 ? 1 : 3)) {
				size_t zl = n + n / 2 + 1024, bl = n;
				if (hdo_deflate_twin(z, &zl, in, n, level)) { fails++; continue; }
				uint64_t bits;
				if (hdo_inflate(back, &bl, z, zl, &bits) || bl != n || memcmp(back, in, n)) fails++;
				/* truncated and corrupted streams must not crash */
				for (int t = 0; t < 4 && zl > 2; t++) {
					size_t cut = rnd() % zl, bl2 = n;
					hdo_inflate(back, &bl2, z, cut, NULL);
					size_t k = rnd() % zl;
					uint8_t old = z[k];
					z[k] ^= (uint8_t)(1u << (rnd() & 7));
					bl2 = n;
					hdo_inflate(back, &bl2, z, zl, NULL);
					z[k] = old;
				}
			}
			uint8_t *m = malloc(70000);
			if (n <= 65000)
				hdo_bgzf_frame(m, 70000, in, n, hdo_crc32(0, in, n), (uint32_t)n);
			hdo_adler32(1, in, n);
			free(m); free(in); free(z); free(back);
		}
	}
	printf("san_harness: %d failures\n", fails);
	return fails != 0;
}
