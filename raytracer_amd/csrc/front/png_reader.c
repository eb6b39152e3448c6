/* PNG reader for cubemap faces: 8-bit RGB / RGBA, non-interlaced, which is the
 * subset the reference accepts (texture.c:31 requires 3 or 4 bytes per pixel;
 * pnglite rejects interlaced and indexed images, pnglite/pnglite.c:190-197).
 * Self-contained: DEFLATE (RFC 1951) is decoded here so the library needs no
 * zlib at run time. */
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "front.h"

/* ----------------------------- inflate ---------------------------------- */

typedef struct BitReader {
	const uint8_t *in;
	size_t len, pos;
	uint32_t bitbuf;
	int bitcnt;
	int error;
} BitReader;

static int getbits(BitReader *br, int n)
{
	while (br->bitcnt < n) {
		if (br->pos >= br->len) {
			br->error = 1;
			return 0;
		}
		br->bitbuf |= (uint32_t)br->in[br->pos++] << br->bitcnt;
		br->bitcnt += 8;
	}
	int v = (int)(br->bitbuf & ((1u << n) - 1u));
	br->bitbuf >>= n;
	br->bitcnt -= n;
	return v;
}

#define MAXBITS 15

typedef struct Huffman {
	uint16_t count[MAXBITS + 1];
	uint16_t symbol[288];
} Huffman;

static int huff_build(Huffman *h, const uint8_t *lengths, int n)
{
	uint16_t offs[MAXBITS + 1];
	memset(h->count, 0, sizeof h->count);
	for (int i = 0; i < n; i++)
		h->count[lengths[i]]++;
	if (h->count[0] == n)
		return 0;
	int left = 1;
	for (int len = 1; len <= MAXBITS; len++) {
		left <<= 1;
		left -= h->count[len];
		if (left < 0)
			return -1;              /* over-subscribed */
	}
	offs[1] = 0;
	for (int len = 1; len < MAXBITS; len++)
		offs[len + 1] = offs[len] + h->count[len];
	for (int i = 0; i < n; i++)
		if (lengths[i])
			h->symbol[offs[lengths[i]]++] = (uint16_t)i;
	return left;
}

static int huff_decode(BitReader *br, const Huffman *h)
{
	int code = 0, first = 0, index = 0;
	for (int len = 1; len <= MAXBITS; len++) {
		code |= getbits(br, 1);
		if (br->error)
			return -1;
		int count = h->count[len];
		if (code - count < first)
			return h->symbol[index + (code - first)];
		index += count;
		first += count;
		first <<= 1;
		code <<= 1;
	}
	return -1;
}

static const uint16_t len_base[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19,
	23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
static const uint16_t len_extra[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2,
	3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
static const uint16_t dist_base[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97,
	129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289,
	16385, 24577 };
static const uint16_t dist_extra[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6,
	7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };

static int inflate_codes(BitReader *br, uint8_t *out, size_t out_len, size_t *opos,
		const Huffman *lencode, const Huffman *distcode)
{
	for (;;) {
		int sym = huff_decode(br, lencode);
		if (sym < 0)
			return -1;
		if (sym < 256) {
			if (*opos >= out_len)
				return -1;
			out[(*opos)++] = (uint8_t)sym;
		} else if (sym == 256) {
			return 0;
		} else {
			sym -= 257;
			if (sym >= 29)
				return -1;
			int len = len_base[sym] + getbits(br, len_extra[sym]);
			int dsym = huff_decode(br, distcode);
			if (dsym < 0 || dsym >= 30)
				return -1;
			size_t dist = (size_t)dist_base[dsym] + (size_t)getbits(br, dist_extra[dsym]);
			if (br->error || dist > *opos || *opos + (size_t)len > out_len)
				return -1;
			for (int i = 0; i < len; i++, (*opos)++)
				out[*opos] = out[*opos - dist];
		}
	}
}

int front_inflate(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len)
{
	if (in_len < 6)
		return RTAMD_ERR_IO;
	/* zlib header: CM = 8, no preset dictionary, FCHECK */
	if ((in[0] & 0x0f) != 8 || (in[1] & 0x20) || ((in[0] << 8 | in[1]) % 31) != 0)
		return RTAMD_ERR_IO;
	BitReader br = { in + 2, in_len - 2, 0, 0, 0, 0 };
	size_t opos = 0;
	int last;
	do {
		last = getbits(&br, 1);
		int type = getbits(&br, 2);
		if (br.error)
			return RTAMD_ERR_IO;
		if (type == 0) {
			br.bitbuf = 0;
			br.bitcnt = 0;
			if (br.pos + 4 > br.len)
				return RTAMD_ERR_IO;
			unsigned len = br.in[br.pos] | (br.in[br.pos + 1] << 8);
			unsigned nlen = br.in[br.pos + 2] | (br.in[br.pos + 3] << 8);
			br.pos += 4;
			if ((len ^ 0xffffu) != nlen || br.pos + len > br.len || opos + len > out_len)
				return RTAMD_ERR_IO;
			memcpy(out + opos, br.in + br.pos, len);
			opos += len;
			br.pos += len;
		} else if (type == 1) {
			Huffman lc, dc;
			uint8_t lengths[288];
			int i;
			for (i = 0; i < 144; i++) lengths[i] = 8;
			for (; i < 256; i++) lengths[i] = 9;
			for (; i < 280; i++) lengths[i] = 7;
			for (; i < 288; i++) lengths[i] = 8;
			huff_build(&lc, lengths, 288);
			for (i = 0; i < 30; i++) lengths[i] = 5;
			huff_build(&dc, lengths, 30);
			if (inflate_codes(&br, out, out_len, &opos, &lc, &dc))
				return RTAMD_ERR_IO;
		} else if (type == 2) {
			static const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4,
				12, 3, 13, 2, 14, 1, 15 };
			Huffman lc, dc;
			uint8_t lengths[320];
			int nlen = getbits(&br, 5) + 257;
			int ndist = getbits(&br, 5) + 1;
			int ncode = getbits(&br, 4) + 4;
			if (br.error || nlen > 286 || ndist > 30)
				return RTAMD_ERR_IO;
			memset(lengths, 0, sizeof lengths);
			for (int i = 0; i < ncode; i++)
				lengths[order[i]] = (uint8_t)getbits(&br, 3);
			if (huff_build(&lc, lengths, 19) != 0)
				return RTAMD_ERR_IO;
			int idx = 0;
			while (idx < nlen + ndist) {
				int sym = huff_decode(&br, &lc);
				if (sym < 0)
					return RTAMD_ERR_IO;
				if (sym < 16) {
					lengths[idx++] = (uint8_t)sym;
				} else {
					int prev = 0, rep;
					if (sym == 16) {
						if (idx == 0)
							return RTAMD_ERR_IO;
						prev = lengths[idx - 1];
						rep = 3 + getbits(&br, 2);
					} else if (sym == 17)
						rep = 3 + getbits(&br, 3);
					else
						rep = 11 + getbits(&br, 7);
					if (idx + rep > nlen + ndist)
						return RTAMD_ERR_IO;
					while (rep--)
						lengths[idx++] = (uint8_t)prev;
				}
			}
			if (lengths[256] == 0)
				return RTAMD_ERR_IO;
			int e = huff_build(&lc, lengths, nlen);
			if (e < 0 || (e > 0 && nlen - lc.count[0] != 1))
				return RTAMD_ERR_IO;
			e = huff_build(&dc, lengths + nlen, ndist);
			if (e < 0 || (e > 0 && ndist - dc.count[0] != 1))
				return RTAMD_ERR_IO;
			if (inflate_codes(&br, out, out_len, &opos, &lc, &dc))
				return RTAMD_ERR_IO;
		} else
			return RTAMD_ERR_IO;
	} while (!last);
	return opos == out_len ? 0 : RTAMD_ERR_IO;
}

/* ------------------------------- PNG ------------------------------------ */

static uint32_t be32(const uint8_t *p)
{
	return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3];
}

static uint32_t crc32_update(uint32_t crc, const uint8_t *p, size_t n)
{
	static uint32_t table[256];
	static int have;
	if (!have) {
		for (uint32_t i = 0; i < 256; i++) {
			uint32_t c = i;
			for (int k = 0; k < 8; k++)
				c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1;
			table[i] = c;
		}
		have = 1;
	}
	for (size_t i = 0; i < n; i++)
		crc = table[(crc ^ p[i]) & 0xff] ^ (crc >> 8);
	return crc;
}

static int paeth(int a, int b, int c)
{
	int p = a + b - c;
	int pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
	if (pa <= pb && pa <= pc)
		return a;
	return pb <= pc ? b : c;
}

int front_png_load(const char *path, int *width, int *height, int *channels,
		uint8_t **pixels)
{
	static const uint8_t sig[8] = { 137, 80, 78, 71, 13, 10, 26, 10 };
	FILE *f = fopen(path, "rb");
	if (!f) {
		rtamd_set_error("Couldn't open file %s", path);   /* texture.c:26 */
		return RTAMD_ERR_IO;
	}
	fseek(f, 0, SEEK_END);
	long flen = ftell(f);
	fseek(f, 0, SEEK_SET);
	uint8_t *file = malloc((size_t)flen + 1);
	if (!file) {
		fclose(f);
		return RTAMD_ERR_NOMEM;
	}
	size_t got = fread(file, 1, (size_t)flen, f);
	fclose(f);

	int rc = RTAMD_ERR_IO;
	uint8_t *idat = NULL, *raw = NULL, *out = NULL;
	size_t idat_len = 0;
	uint32_t w = 0, h = 0;
	int bpp = 0, seen_ihdr = 0, seen_iend = 0;

	if (got < 8 || memcmp(file, sig, 8) != 0) {
		rtamd_set_error("%s is not a PNG file", path);
		goto done;
	}
	idat = malloc(got);
	if (!idat) {
		rc = RTAMD_ERR_NOMEM;
		goto done;
	}
	for (size_t pos = 8; pos + 12 <= got && !seen_iend;) {
		uint32_t len = be32(file + pos);
		const uint8_t *type = file + pos + 4;
		const uint8_t *data = file + pos + 8;
		if (pos + 12 + (size_t)len > got) {
			rtamd_set_error("%s: truncated PNG chunk", path);
			goto done;
		}
		uint32_t crc = crc32_update(0xffffffffu, type, 4 + (size_t)len) ^ 0xffffffffu;
		if (crc != be32(data + len)) {
			rtamd_set_error("%s: PNG chunk CRC mismatch", path);
			goto done;
		}
		if (memcmp(type, "IHDR", 4) == 0) {
			if (len != 13)
				goto done;
			w = be32(data);
			h = be32(data + 4);
			int depth = data[8], ctype = data[9], interlace = data[12];
			if (depth != 8 || (ctype != 2 && ctype != 6) || interlace != 0 ||
					data[10] != 0 || data[11] != 0) {
				rtamd_set_error("%s: unsupported PNG format (need 8-bit RGB/RGBA, "
						"non-interlaced)", path);
				goto done;
			}
			bpp = ctype == 2 ? 3 : 4;
			seen_ihdr = 1;
		} else if (memcmp(type, "IDAT", 4) == 0) {
			memcpy(idat + idat_len, data, len);
			idat_len += len;
		} else if (memcmp(type, "IEND", 4) == 0) {
			seen_iend = 1;
		}
		pos += 12 + (size_t)len;
	}
	if (!seen_ihdr || !w || !h || w > 16384 || h > 16384) {
		rtamd_set_error("%s: bad PNG header", path);
		goto done;
	}
	size_t stride = (size_t)w * (size_t)bpp;
	size_t raw_len = (stride + 1) * h;
	raw = malloc(raw_len);
	out = malloc(stride * h);
	if (!raw || !out) {
		rc = RTAMD_ERR_NOMEM;
		goto done;
	}
	if (front_inflate(idat, idat_len, raw, raw_len) != 0) {
		rtamd_set_error("Error getting data from png %s", path);   /* texture.c:39 */
		goto done;
	}
	for (uint32_t y = 0; y < h; y++) {
		const uint8_t *src = raw + (stride + 1) * y;
		uint8_t *dst = out + stride * y;
		const uint8_t *up = y ? dst - stride : NULL;
		int filter = src[0];
		src++;
		for (size_t x = 0; x < stride; x++) {
			int a = x >= (size_t)bpp ? dst[x - bpp] : 0;
			int b = up ? up[x] : 0;
			int c = (up && x >= (size_t)bpp) ? up[x - bpp] : 0;
			int v = src[x];
			switch (filter) {
			case 0: break;
			case 1: v += a; break;
			case 2: v += b; break;
			case 3: v += (a + b) >> 1; break;
			case 4: v += paeth(a, b, c); break;
			default:
				rtamd_set_error("%s: bad PNG filter type", path);
				goto done;
			}
			dst[x] = (uint8_t)v;
		}
	}
	*width = (int)w;
	*height = (int)h;
	*channels = bpp;
	*pixels = out;
	out = NULL;
	rc = 0;
done:
	free(file);
	free(idat);
	free(raw);
	free(out);
	return rc;
}
