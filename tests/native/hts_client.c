/* hts_client.c -- the "samtools" of the LD_PRELOAD test: links libfakehts.so only,
 * reads stdin, writes BGZF to stdout with N worker threads (argv[1], default 8). */
#include <stdio.h>
#include <stdlib.h>

int fakehts_write_bgzf(FILE *out, const unsigned char *data, size_t n, int nthreads);

int main(int argc, char **argv)
{
	size_t cap = 1 << 20, n = 0;
	unsigned char *buf = malloc(cap);
	for (;;) {
		size_t got = fread(buf + n, 1, cap - n, stdin);
		n += got;
		if (got == 0)
			break;
		if (n == cap)
			buf = realloc(buf, cap *= 2);
	}
	return fakehts_write_bgzf(stdout, buf, n, argc > 1 ? atoi(argv[1]) : 8);
}
