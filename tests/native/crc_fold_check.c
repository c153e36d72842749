/* CPU check of the container hosts' CRC folding (7bgzf_amd/csrc/hd_host_util.h): reads
 * "<crc_hex> <len>" pairs from stdin, prints the CRC-32 of the concatenation. */
#include <stdio.h>
#include <stdlib.h>
#include "hd_host_util.h"

int main(void)
{
	struct crc_fold f = { 0, 0, 0 };
	unsigned crc;
	unsigned long len;
	while (scanf("%x %lu", &crc, &len) == 2)
		crc_append(&f, crc, (uint32_t)len);
	printf("%08x\n", f.crc);
	return 0;
}
