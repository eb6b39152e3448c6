/*
 * rayviewer -- the progressive caller: the place of the reference's rayviewer.c,
 * without its SDL-1.2 window.
 *
 *   rayviewer scene.sdl [--passes N | --progressive P [--aa A]] [--dump PREFIX] [--assets DIR]
 *             [--device D] [-o out.ppm]
 *
 * The reference's viewer renders the pixels one at a time in a shuffled order, no
 * antialiasing (camera_ray, rayviewer.c:137), and flips its window once per `width`
 * pixels so that the picture fills in evenly (rayviewer.c:77-89,128-150); at the
 * end it writes ray.ppm like the batch driver (rayviewer.c:158-161).  Here one
 * refresh is one rtamd_render call on a subset of the 32x32 screen tiles: pass k
 * of N renders the tiles k, k+N, k+2N, ... (the same interleave the multi-GPU
 * partition uses), the passes run in a shuffled order, and after each pass the
 * frame is read back and quantised for display the way put_pixel does it
 * (rayviewer.c:62-75: floor(c * 256) clamped to 0..255, top row first).  A window
 * system would blit that surface; --dump writes it to PREFIX-<refresh>.ppm instead.
 * The finished frame does not depend on N or on the order of the passes.
 *
 * --progressive P refines by SAMPLES instead: antialiasing stays on (the scene's own
 * aa_samples, or --aa A) and every pass renders a share of each pixel's samples
 * (RTAMD_RENDER_PROGRESSIVE), so already the first refresh shows the whole picture,
 * noisy, and every further one adds samples to all pixels.  Sample counts that divide
 * 64 are taken in order, in P equal runs (P rounded down to a power of two): the
 * library adds them in float in the reference's order.  Other counts are dealt out
 * k, k+P, k+2P, ... in a shuffled order of the passes.  The last refresh is the batch
 * driver's antialiased frame, bit for bit.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "rtamd.h"

static double now(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static int fail(const char *what)
{
	printf("%s\n", what);
	return 1;
}

/* put_pixel, rayviewer.c:62-75 */
static void to_display(const float *colour, int w, int h, unsigned char *surface)
{
	for (int y = 0; y < h; y++)
		for (int x = 0; x < w; x++) {
			const float *c = colour + 4 * ((size_t)w * (h - 1 - y) + x);
			unsigned char *p = surface + 3 * ((size_t)w * y + x);
			for (int k = 0; k < 3; k++) {
				float v = floorf(c[k] * 256);
				p[k] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
			}
		}
}

int main(int argc, char **argv)
{
	if (argc < 2)
		return 1;                                 /* rayviewer.c:101-102 */
	const char *sdl = argv[1], *assets = NULL, *out = "ray.ppm", *dump = NULL;
	int passes = 16, device = 0, progressive = 0, aa = -1;
	for (int a = 2; a < argc; a++) {
		if (!strcmp(argv[a], "--passes") && a + 1 < argc) passes = atoi(argv[++a]);
		else if (!strcmp(argv[a], "--progressive") && a + 1 < argc) { progressive = 1; passes = atoi(argv[++a]); }
		else if (!strcmp(argv[a], "--aa") && a + 1 < argc) aa = atoi(argv[++a]);
		else if (!strcmp(argv[a], "--dump") && a + 1 < argc) dump = argv[++a];
		else if (!strcmp(argv[a], "--assets") && a + 1 < argc) assets = argv[++a];
		else if (!strcmp(argv[a], "--device") && a + 1 < argc) device = atoi(argv[++a]);
		else if (!strcmp(argv[a], "-o") && a + 1 < argc) out = argv[++a];
		else {
			printf("Unknown option %s\n", argv[a]);
			return 1;
		}
	}
	void *blob = NULL;
	size_t blob_size = 0;
	if (rtamd_sdl_load(sdl, assets, &blob, &blob_size))
		return fail(rtamd_last_error());
	rtamd_blob_header *h = blob;
	if (!progressive)
		h->antialiasing = 0;                      /* rayviewer.c:137: always camera_ray */
	else if (aa > 0) {
		h->antialiasing = 1;
		h->aa_samples = aa;
	}
	const int spp = h->antialiasing ? h->aa_samples * h->aa_samples : 1;
	const int ordered = progressive && 64 % spp == 0;
	if (progressive && passes > spp)
		passes = spp;
	if (ordered) {
		int p2 = 1;
		while (2 * p2 <= passes)
			p2 *= 2;
		passes = p2;
	}
	const int w = h->width, hh = h->height;
	rtamd_scene *scene = NULL;
	if (rtamd_init(device) || rtamd_scene_create(blob, blob_size, &scene))
		return fail(rtamd_last_error());
	const size_t bytes = (size_t)w * hh * 4 * sizeof(float);
	void *d_colour = NULL;
	float *colour = malloc(bytes);
	unsigned char *surface = malloc((size_t)w * hh * 3);
	if (!colour || !surface || hipMalloc(&d_colour, bytes) != hipSuccess || hipMemset(d_colour, 0, bytes) != hipSuccess)
		return fail("Could not allocate the frame buffer");

	if (passes < 1) passes = 1;
	int *order = malloc(sizeof(int) * (size_t)passes);
	for (int i = 0; i < passes; i++)
		order[i] = i;
	srand((unsigned)time(NULL));                  /* rayviewer.c:121-122: the order differs from run to run */
	for (int i = passes - 1; i >= 1; i--) {
		int j = rand() % (i + 1), t = order[i];
		order[i] = order[j];
		order[j] = t;
	}

	rtamd_render_params p;
	memset(&p, 0, sizeof p);
	p.seed = 0x20071208;                          /* rayviewer.c:123 */
	p.rng_mode = RTAMD_RNG_KEYED;
	p.tile_w = p.tile_h = 32;
	p.tile_stride = progressive ? 1 : passes;
	const double t0 = now();
	for (int i = 0; i < passes; i++) {
		if (ordered) {
			p.flags = RTAMD_RENDER_PROGRESSIVE | (i == 0 ? RTAMD_RENDER_RESTART : 0);
			p.sample_count = spp / passes;
			p.sample_first = i * p.sample_count;
			p.sample_stride = 1;
		} else if (progressive) {
			p.flags = RTAMD_RENDER_PROGRESSIVE | (i == 0 ? RTAMD_RENDER_RESTART : 0);
			p.sample_first = order[i];
			p.sample_stride = passes;
		} else
			p.tile_first = order[i];
		if (rtamd_render(scene, &p, d_colour, NULL, NULL, NULL))
			return fail(rtamd_last_error());
		/* the refresh (SDL_Flip, rayviewer.c:143-149) */
		if (hipMemcpy(colour, d_colour, bytes, hipMemcpyDeviceToHost) != hipSuccess)
			return fail("Could not read the frame buffer back");
		to_display(colour, w, hh, surface);
		if (dump) {
			char name[4096];
			snprintf(name, sizeof name, "%s-%03d.ppm", dump, i);
			FILE *f = fopen(name, "wb");
			if (!f)
				return fail("Could not write a refresh");
			fprintf(f, "P6\n%d %d\n255\n", w, hh);
			fwrite(surface, 3, (size_t)w * hh, f);
			fclose(f);
		}
	}
	const double seconds = now() - t0;
	int sec = (int)seconds, msec = (int)((seconds - sec) * 1000);
	printf("Rendering took %d s %03d msec\n", sec, msec);                      /* timer.c:48-57 */
	printf("%.2f kilopixels per second\n", w * hh / 1000. / seconds);        /* rayviewer.c:155-156 */
	if (rtamd_ppm_write(colour, w, hh, out))
		return fail(rtamd_last_error());
	free(order);
	free(surface);
	free(colour);
	(void)hipFree(d_colour);
	rtamd_scene_destroy(scene);
	rtamd_blob_free(blob);
	rtamd_shutdown();
	return 0;
}
