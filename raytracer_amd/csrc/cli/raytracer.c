/*
 * raytracer -- host driver in C: the place of the reference's raytracer.c.
 *
 *   raytracer scene.sdl                       (the reference's one-argument form)
 *   raytracer scene.sdl [--width W] [--height H] [--aa N | --no-aa] [--seed S]
 *             [--rng glibc|keyed] [--device D | --devices D0,D1,...] [--bands N]
 *             [--assets DIR] [-o out.ppm | -o out.png] [--stats]
 *
 * Console contract of the reference, line for line: "Loading cubemap took ..."
 * when the scene has one (scene.c:570-573), per mesh "Building kd-tree for
 * <shape name>" and "Building kd-tree took ..." (scene.c:662-666; here the BVH
 * build), the 70-column progress bar redrawn with '\r' and closed by a newline
 * (raytracer.c:13-30,83,85), "Rendering took S s MMM msec", "%.2f kilopixels
 * per second" (raytracer.c:89-92), and the image in ray.ppm in the current
 * directory.  --stats adds one line with ray counts, which the reference does
 * not print.
 *
 * The pixel loop raytracer.c:78-84 becomes rtamd_render calls: the frame is
 * rendered in horizontal bands of whole tile rows (a pixel window per call), the
 * bar advancing after each band as the reference's does after each row.
 * --devices renders with several GPUs from this one process: every device gets
 * the tiles dealt to it cyclically, and the other devices' packed tiles are
 * copied peer to peer into device 0's frame.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "rtamd.h"

#define MAX_DEVICES 16
#define TILE 32

static double now(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* timer_diff_print, timer.c:48-57 */
static void print_took(const char *name, double seconds)
{
	int sec = (int)seconds;
	int msec = (int)((seconds - sec) * 1000);
	printf("%s took %d s %03d msec\n", name, sec, msec);
}

/* raytracer.c:13-30 */
static void print_progressbar(int progress, int total)
{
	const int tot_bars = 70;
	int bars = total > 0 ? (int)(((long long)progress * tot_bars) / total) : tot_bars;
	printf("\r[");
	for (int i = 0; i < bars; i++)
		printf("=");
	printf(">");
	for (int i = bars + 1; i <= tot_bars; i++)
		printf(" ");
	printf("]\r");
	fflush(stdout);
}

static int fail(const char *what)
{
	printf("%s\n", what);
	return 1;
}

int main(int argc, char **argv)
{
	if (argc < 2)
		return 1;                                 /* raytracer.c:63-64 */
	const char *sdl = argv[1], *assets = NULL, *out = "ray.ppm";
	int width = 0, height = 0, aa = -1, rng = -1, bands = 70, want_stats = 0;
	int devices[MAX_DEVICES] = { 0 }, ndev = 1;
	unsigned seed = 0x20071208;                   /* raytracer.c:74 */
	for (int a = 2; a < argc; a++) {
		if (!strcmp(argv[a], "--width") && a + 1 < argc) width = atoi(argv[++a]);
		else if (!strcmp(argv[a], "--height") && a + 1 < argc) height = atoi(argv[++a]);
		else if (!strcmp(argv[a], "--aa") && a + 1 < argc) aa = atoi(argv[++a]);
		else if (!strcmp(argv[a], "--no-aa")) aa = 0;
		else if (!strcmp(argv[a], "--seed") && a + 1 < argc) seed = (unsigned)strtoul(argv[++a], NULL, 0);
		else if (!strcmp(argv[a], "--device") && a + 1 < argc) { devices[0] = atoi(argv[++a]); ndev = 1; }
		else if (!strcmp(argv[a], "--devices") && a + 1 < argc) {
			ndev = 0;
			for (char *tok = strtok(argv[++a], ","); tok && ndev < MAX_DEVICES; tok = strtok(NULL, ","))
				devices[ndev++] = atoi(tok);
			if (ndev == 0)
				return fail("--devices needs a comma-separated list of device numbers");
		}
		else if (!strcmp(argv[a], "--bands") && a + 1 < argc) bands = atoi(argv[++a]);
		else if (!strcmp(argv[a], "--assets") && a + 1 < argc) assets = argv[++a];
		else if (!strcmp(argv[a], "--rng") && a + 1 < argc) rng = !strcmp(argv[++a], "keyed");
		else if (!strcmp(argv[a], "--stats")) want_stats = 1;
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
	if (rtamd_last_cubemap_load_seconds() >= 0.0)
		print_took("Loading cubemap", rtamd_last_cubemap_load_seconds());
	rtamd_blob_header *h = blob;
	if (width > 0) h->width = width;
	if (height > 0) h->height = height;
	if (aa == 0) h->antialiasing = 0;
	if (aa > 0) { h->antialiasing = 1; h->aa_samples = aa; }

	const int w = h->width, hh = h->height;
	const size_t bytes = (size_t)w * hh * 4 * sizeof(float);
	rtamd_scene *scene[MAX_DEVICES] = { NULL };
	void *d_colour[MAX_DEVICES] = { NULL };
	for (int k = 0; k < ndev; k++) {
		if (rtamd_init(devices[k]) || rtamd_scene_create(blob, blob_size, &scene[k]))
			return fail(rtamd_last_error());
		if (hipMalloc(&d_colour[k], bytes) != hipSuccess || hipMemset(d_colour[k], 0, bytes) != hipSuccess)
			return fail("Could not allocate the frame buffer on the device");
	}
	/* scene.c:655-668: one kd-tree per mesh, built when the surface walk first meets it */
	{
		const rtamd_surface *surf = (const rtamd_surface *)((const char *)blob + h->off_surfaces);
		const rtamd_mesh *mesh = (const rtamd_mesh *)((const char *)blob + h->off_meshes);
		char *seen = calloc((size_t)h->num_meshes + 1, 1);
		for (int s = 0; s < h->num_surfaces; s++) {
			const int m = surf[s].mesh;
			if (m < 0 || seen[m])
				continue;
			seen[m] = 1;
			double secs = 0.0;
			rtamd_scene_mesh_build_seconds(scene[0], m, &secs);
			printf("Building kd-tree for %s\n", mesh[m].off_name ? (const char *)blob + mesh[m].off_name : "");
			print_took("Building kd-tree", secs);
		}
		free(seen);
	}

	rtamd_render_params p;
	memset(&p, 0, sizeof p);
	/* tile exchange buffers of a multi-device run, allocated before the clock starts: a
	 * dense block of owner 0's size (the largest) per device, and their landing place on
	 * the first device */
	void *packed[MAX_DEVICES] = { NULL }, *landing = NULL;
	size_t pbytes = 0;
	if (ndev > 1) {
		p.tile_w = p.tile_h = TILE;
		p.tile_stride = ndev;
		pbytes = (size_t)rtamd_tiles_owned(w, hh, &p) * TILE * TILE * 4 * sizeof(float);
		if (hipSetDevice(devices[0]) != hipSuccess || hipMalloc(&landing, pbytes * ndev) != hipSuccess)
			return fail("Could not allocate the tile exchange buffer");
		for (int k = 1; k < ndev; k++)
			if (hipSetDevice(devices[k]) != hipSuccess || hipMalloc(&packed[k], pbytes) != hipSuccess ||
					hipMemset(packed[k], 0, pbytes) != hipSuccess)
				return fail("Could not allocate the tile exchange buffer");
	}
	p.seed = seed;
	p.rng_mode = rng >= 0 ? rng : RTAMD_RNG_GLIBC;
	p.tile_w = p.tile_h = TILE;
	p.tile_stride = ndev;
	const int tile_rows = (hh + TILE - 1) / TILE;
	if (bands < 1) bands = 1;
	if (bands > tile_rows) bands = tile_rows;
	const int rows_per_band = (tile_rows + bands - 1) / bands * TILE;
	unsigned long long rays = 0;
	double device_ms = 0.0;
	const double t0 = now();
	for (int y0 = 0; y0 < hh; y0 += rows_per_band) {
		const int y1 = y0 + rows_per_band < hh ? y0 + rows_per_band : hh;
		p.x0 = 0; p.x1 = w; p.y0 = y0; p.y1 = y1;
		for (int k = 0; k < ndev; k++) {
			rtamd_render_stats st;
			p.tile_first = k;
			int rc = rtamd_render(scene[k], &p, d_colour[k], NULL, want_stats ? &st : NULL, NULL);
			if (rc == RTAMD_ERR_ARG && rng < 0 && p.rng_mode == RTAMD_RNG_GLIBC) {
				/* the serial stream cannot be replayed for this scene: keyed samples */
				p.rng_mode = RTAMD_RNG_KEYED;
				rc = rtamd_render(scene[k], &p, d_colour[k], NULL, want_stats ? &st : NULL, NULL);
			}
			if (rc)
				return fail(rtamd_last_error());
			if (want_stats) {
				rays += st.rays_primary + st.rays_shadow + st.rays_secondary;
				device_ms += st.ms_total;
			}
		}
		for (int k = 0; k < ndev; k++)
			if (hipSetDevice(devices[k]) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
				return fail("A device failed while rendering");
		print_progressbar(y1 - 1, hh - 1);        /* raytracer.c:83 */
	}
	/* the other devices' tiles: packed there, copied peer to peer into device 0's landing
	 * buffer (one block per device) by asynchronous copies that run side by side, then
	 * unpacked into device 0's frame by one launch */
	if (ndev > 1) {
		p.x0 = p.y0 = p.x1 = p.y1 = 0;
		for (int k = 1; k < ndev; k++) {
			p.tile_first = k;
			if (hipSetDevice(devices[k]) != hipSuccess)
				return fail("Could not select a device");
			if (rtamd_tiles_pack(w, hh, &p, d_colour[k], packed[k], NULL))
				return fail(rtamd_last_error());
			/* on device k's null stream, behind the pack kernel */
			if (hipMemcpyPeerAsync((char *)landing + (size_t)k * pbytes, devices[0], packed[k], devices[k], pbytes,
					NULL) != hipSuccess)
				return fail("Peer copy of the rendered tiles failed");
		}
		for (int k = 1; k < ndev; k++)
			if (hipSetDevice(devices[k]) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
				return fail("Peer copy of the rendered tiles failed");
		p.tile_first = 0;
		if (hipSetDevice(devices[0]) != hipSuccess ||
				rtamd_tiles_unpack_all(w, hh, &p, 1, landing, pbytes, d_colour[0], NULL))
			return fail(rtamd_last_error());
		if (hipDeviceSynchronize() != hipSuccess)
			return fail("Unpacking the rendered tiles failed");
	}
	const double seconds = now() - t0;
	printf("\n");                                 /* raytracer.c:85 */

	float *colour = malloc(bytes);
	if (hipSetDevice(devices[0]) != hipSuccess || !colour ||
			hipMemcpy(colour, d_colour[0], bytes, hipMemcpyDeviceToHost) != hipSuccess)
		return fail("Could not read the frame buffer back");
	print_took("Rendering", seconds);
	printf("%.2f kilopixels per second\n", w * hh / 1000. / seconds);
	if (want_stats)
		printf("%llu rays, %.3f ms on the device%s, %.1f Mrays/s\n", rays, device_ms, ndev > 1 ? "s" : "",
				rays / (device_ms * 1e-3) / 1e6);
	/* ray.ppm as the reference writes it; a name ending in .png selects the PNG writer */
	const size_t len = strlen(out);
	const int png = len > 4 && !strcmp(out + len - 4, ".png");
	if (png ? rtamd_png_write(colour, w, hh, out) : rtamd_ppm_write(colour, w, hh, out))
		return fail(rtamd_last_error());
	free(colour);
	for (int k = 0; k < ndev; k++) {
		(void)hipSetDevice(devices[k]);
		(void)hipFree(d_colour[k]);
		if (packed[k])
			(void)hipFree(packed[k]);
		if (k == 0 && landing)
			(void)hipFree(landing);
		rtamd_scene_destroy(scene[k]);
	}
	rtamd_blob_free(blob);
	rtamd_shutdown();
	return 0;
}
