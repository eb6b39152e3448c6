/*
 * raytracer -- host driver in C: the place of the reference's raytracer.c.
 *
 *   raytracer scene.sdl                       (the reference's one-argument form)
 *   raytracer scene.sdl [--width W] [--height H] [--aa N | --no-aa] [--seed S]
 *             [--rng glibc|keyed] [--device D] [--assets DIR] [-o out.ppm]
 *
 * Same console contract as the reference (raytracer.c:85-92, scene.c:570-573,
 * 662-666): the "Building kd-tree" lines (here: the BVH build), "Rendering took
 * S s MMM msec", "%.2f kilopixels per second", and the image in ray.ppm in the
 * current directory.  The pixel loop raytracer.c:78-84 is one rtamd_render call.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

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

/* timer_diff_print, timer.c:48-57 */
static void print_took(const char *name, double seconds)
{
	long sec = (long)seconds;
	long msec = (long)((seconds - (double)sec) * 1000.0);
	printf("%s took %ld s %03ld msec\n", name, sec, msec);
}

int main(int argc, char **argv)
{
	if (argc < 2)
		return 1;                                 /* raytracer.c:63-64 */
	const char *sdl = argv[1], *assets = NULL, *out = "ray.ppm";
	int width = 0, height = 0, aa = -1, device = 0, rng = -1;
	unsigned seed = 0x20071208;                   /* raytracer.c:74 */
	for (int a = 2; a < argc; a++) {
		if (!strcmp(argv[a], "--width") && a + 1 < argc) width = atoi(argv[++a]);
		else if (!strcmp(argv[a], "--height") && a + 1 < argc) height = atoi(argv[++a]);
		else if (!strcmp(argv[a], "--aa") && a + 1 < argc) aa = atoi(argv[++a]);
		else if (!strcmp(argv[a], "--no-aa")) aa = 0;
		else if (!strcmp(argv[a], "--seed") && a + 1 < argc) seed = (unsigned)strtoul(argv[++a], NULL, 0);
		else if (!strcmp(argv[a], "--device") && a + 1 < argc) device = atoi(argv[++a]);
		else if (!strcmp(argv[a], "--assets") && a + 1 < argc) assets = argv[++a];
		else if (!strcmp(argv[a], "--rng") && a + 1 < argc) rng = !strcmp(argv[++a], "keyed");
		else if (!strcmp(argv[a], "-o") && a + 1 < argc) out = argv[++a];
		else {
			printf("Unknown option %s\n", argv[a]);
			return 1;
		}
	}

	void *blob = NULL;
	size_t blob_size = 0;
	if (rtamd_sdl_load(sdl, assets, &blob, &blob_size)) {
		printf("%s\n", rtamd_last_error());
		printf("Error importing SDL from file %s\n", sdl);   /* scene.c:692 */
		return 1;                                 /* raytracer.c:67-68 */
	}
	rtamd_blob_header *h = blob;
	if (width > 0) h->width = width;
	if (height > 0) h->height = height;
	if (aa == 0) h->antialiasing = 0;
	if (aa > 0) { h->antialiasing = 1; h->aa_samples = aa; }

	if (rtamd_init(device)) {
		printf("%s\n", rtamd_last_error());
		return 1;
	}
	rtamd_scene *scene = NULL;
	double t0 = now();
	if (h->num_meshes)
		printf("Building kd-tree for %d mesh%s\n", h->num_meshes, h->num_meshes == 1 ? "" : "es");
	if (rtamd_scene_create(blob, blob_size, &scene)) {
		printf("%s\n", rtamd_last_error());
		return 1;
	}
	if (h->num_meshes)
		print_took("Building kd-tree", now() - t0);

	const int w = h->width, hh = h->height;
	size_t bytes = (size_t)w * hh * 4 * sizeof(float);
	void *d_colour = NULL;
	if (hipMalloc(&d_colour, bytes) != hipSuccess || hipMemset(d_colour, 0, bytes) != hipSuccess) {
		printf("Could not allocate the frame buffer on the device\n");
		return 1;
	}
	rtamd_render_params p;
	memset(&p, 0, sizeof p);
	p.seed = seed;
	p.rng_mode = rng >= 0 ? rng : RTAMD_RNG_GLIBC;
	rtamd_render_stats st;
	t0 = now();
	int rc = rtamd_render(scene, &p, d_colour, NULL, &st, NULL);
	if (rc == RTAMD_ERR_ARG && rng < 0) {
		/* the serial stream cannot be replayed for this scene: keyed samples */
		p.rng_mode = RTAMD_RNG_KEYED;
		t0 = now();
		rc = rtamd_render(scene, &p, d_colour, NULL, &st, NULL);
	}
	if (rc) {
		printf("%s\n", rtamd_last_error());
		return 1;
	}
	double seconds = now() - t0;
	float *colour = malloc(bytes);
	if (!colour || hipMemcpy(colour, d_colour, bytes, hipMemcpyDeviceToHost) != hipSuccess) {
		printf("Could not read the frame buffer back\n");
		return 1;
	}
	print_took("Rendering", seconds);
	printf("%.2f kilopixels per second\n", w * hh / 1000. / seconds);
	unsigned long long rays = st.rays_primary + st.rays_shadow + st.rays_secondary;
	printf("%llu rays, %.3f ms on the device, %.1f Mrays/s\n", rays, st.ms_total,
			rays / (st.ms_total * 1e-3) / 1e6);
	if (rtamd_ppm_write(colour, w, hh, out)) {
		printf("%s\n", rtamd_last_error());
		return 1;
	}
	free(colour);
	(void)hipFree(d_colour);
	rtamd_scene_destroy(scene);
	rtamd_blob_free(blob);
	rtamd_shutdown();
	return 0;
}
