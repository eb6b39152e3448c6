/* PNG output: 8-bit RGB, non-interlaced, from the reference's frame layout with
 * the reference's quantiser (ppm.c:20-24, through rtamd_quantise_rgb8).  The
 * reference never writes PNG; its vendored pnglite has a writer nobody calls
 * (pnglite.c:822-846), and this stands where a caller of it would.  The image
 * data is a zlib stream of stored (uncompressed) deflate blocks, filter type 0
 * on every scanline: valid for every decoder, no compressor needed. */
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "front.h"

static uint32_t crc_table[256];

static void crc_init(void)
{
	if (crc_table[1])
		return;
	for (uint32_t n = 0; n < 256; n++) {
		uint32_t c = n;
		for (int k = 0; k < 8; k++)
			c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1;
		crc_table[n] = c;
	}
}

static uint32_t crc_update(uint32_t crc, const uint8_t *p, size_t n)
{
	for (size_t i = 0; i < n; i++)
		crc = crc_table[(crc ^ p[i]) & 0xff] ^ (crc >> 8);
	return crc;
}

static void put32(uint8_t *p, uint32_t v)
{
	p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v;
}

static int write_chunk(FILE *f, const char type[4], const uint8_t *data, size_t n)
{
	uint8_t head[8], tail[4];
	put32(head, (uint32_t)n);
	memcpy(head + 4, type, 4);
	uint32_t crc = crc_update(0xffffffffu, head + 4, 4);
	crc = crc_update(crc, data, n);
	put32(tail, crc ^ 0xffffffffu);
	return fwrite(head, 1, 8, f) == 8 && fwrite(data, 1, n, f) == n && fwrite(tail, 1, 4, f) == 4;
}

int rtamd_png_write(const float *colour, int width, int height, const char *path)
{
	if (!colour || !path || width <= 0 || height <= 0) {
		rtamd_set_error("rtamd_png_write: bad argument");
		return RTAMD_ERR_ARG;
	}
	crc_init();
	const size_t row = (size_t)3 * (size_t)width, raw_len = (row + 1) * (size_t)height;
	if (raw_len > 0x7fffffffu - 65536u) {
		rtamd_set_error("rtamd_png_write: image too large for one IDAT chunk");
		return RTAMD_ERR_LIMIT;
	}
	uint8_t *rgb = malloc(row * (size_t)height);
	const size_t blocks = (raw_len + 65534) / 65535;
	const size_t z_len = 2 + raw_len + 5 * blocks + 4;
	uint8_t *z = malloc(z_len);
	if (!rgb || !z) {
		free(rgb);
		free(z);
		return RTAMD_ERR_NOMEM;
	}
	rtamd_quantise_rgb8(colour, width, height, rgb);
	/* zlib header, stored blocks over the filtered scanlines, Adler-32 */
	size_t zp = 0;
	z[zp++] = 0x78; z[zp++] = 0x01;
	uint32_t a = 1, b = 0;
	size_t produced = 0, in_block = 0;
	for (int j = 0; j < height; j++) {
		for (size_t k = 0; k <= row; k++) {
			const uint8_t v = k == 0 ? 0 : rgb[row * (size_t)j + k - 1];   /* filter byte 0, then the pixels */
			if (in_block == 0) {
				const size_t left = raw_len - produced;
				const uint32_t len = (uint32_t)(left < 65535 ? left : 65535);
				z[zp++] = left <= 65535 ? 1 : 0;                           /* BFINAL, BTYPE = 00 */
				z[zp++] = (uint8_t)len; z[zp++] = (uint8_t)(len >> 8);
				z[zp++] = (uint8_t)~len; z[zp++] = (uint8_t)(~len >> 8);
				in_block = len;
			}
			z[zp++] = v;
			in_block--;
			produced++;
			a += v; if (a >= 65521) a -= 65521;
			b += a; if (b >= 65521) b -= 65521;
		}
	}
	put32(z + zp, (b << 16) | a);
	zp += 4;
	free(rgb);

	FILE *f = fopen(path, "wb");
	if (!f) {
		rtamd_set_error("Cannot open %s for writing: %s", path, strerror(errno));
		free(z);
		return RTAMD_ERR_IO;
	}
	static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
	uint8_t ihdr[13];
	put32(ihdr, (uint32_t)width);
	put32(ihdr + 4, (uint32_t)height);
	ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;    /* 8-bit RGB, no interlace */
	int ok = fwrite(sig, 1, 8, f) == 8 && write_chunk(f, "IHDR", ihdr, 13) && write_chunk(f, "IDAT", z, zp) &&
			write_chunk(f, "IEND", NULL, 0);
	free(z);
	if (fclose(f) != 0 || !ok) {
		rtamd_set_error("PNG write failed");
		return RTAMD_ERR_IO;
	}
	return 0;
}
