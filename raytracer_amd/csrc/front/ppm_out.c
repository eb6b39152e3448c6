/* PPM output with the reference's 8-bit quantiser (ppm.c:8-35): rows are
 * written top first from a bottom-up buffer, and each channel is
 * (unsigned char)CLAMP(c*255, 0, 255), i.e. truncation, not rounding. */
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "front.h"

static unsigned char quantise(float c)
{
	float x = c * 255;                       /* float * int -> float, ppm.c:22 */
	/* CLAMP(x, 0, 255) then conversion to unsigned char (cgmath.h:14).  A NaN
	 * fails both comparisons and is converted as is; that conversion is
	 * undefined in C and is pinned to 0 here. */
	if (x < 0)
		return 0;
	if (x > 255)
		return 255;
	if (x != x)
		return 0;
	return (unsigned char)x;
}

int rtamd_quantise_rgb8(const float *colour, int width, int height, uint8_t *rgb8)
{
	if (!colour || !rgb8 || width <= 0 || height <= 0) {
		rtamd_set_error("rtamd_quantise_rgb8: bad argument");
		return RTAMD_ERR_ARG;
	}
	for (int j = 0; j < height; j++)
		for (int i = 0; i < width; i++) {
			const float *c = colour + 4 * ((size_t)width * (size_t)(height - 1 - j) + (size_t)i);
			uint8_t *o = rgb8 + 3 * ((size_t)width * (size_t)j + (size_t)i);
			o[0] = quantise(c[0]);
			o[1] = quantise(c[1]);
			o[2] = quantise(c[2]);
		}
	return 0;
}

int rtamd_ppm_write(const float *colour, int width, int height, const char *path)
{
	if (!colour || !path || width <= 0 || height <= 0) {
		rtamd_set_error("rtamd_ppm_write: bad argument");
		return RTAMD_ERR_ARG;
	}
	uint8_t *rgb = malloc((size_t)3 * (size_t)width * (size_t)height);
	if (!rgb)
		return RTAMD_ERR_NOMEM;
	rtamd_quantise_rgb8(colour, width, height, rgb);
	FILE *fd = fopen(path, "wb");
	if (!fd) {
		rtamd_set_error("Cannot open %s for writing: %s", path, strerror(errno));
		free(rgb);
		return RTAMD_ERR_IO;
	}
	fprintf(fd, "P6\n%d %d\n%d\n", width, height, 255);
	size_t want = (size_t)3 * (size_t)width * (size_t)height;
	size_t put = fwrite(rgb, 1, want, fd);
	free(rgb);
	if (fclose(fd) != 0 || put != want) {
		rtamd_set_error("PPM write failed");       /* ppm.c:28 */
		return RTAMD_ERR_IO;
	}
	return 0;
}
