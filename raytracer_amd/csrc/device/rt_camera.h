// Sample index -> pixel (tiles), camera rays (ray.c:19-65), sample offsets.
//
// Part of the device code of kernels.hip, which includes it inside its anonymous
// namespace (after <hip/hip_runtime.h>, scene_dev.h and the parts before it).

// Sample index -> pixel.  Samples are numbered tile by tile over the tiles this
// call owns, pixels row-major inside a tile, `spp` consecutive samples per pixel.
struct TileMap {
	int tile_w, tile_h, tiles_x, tiles_y;
	int tile_first, tile_stride, owned;
	int x0, y0, x1, y1;
};

DEV bool sample_to_pixel(const TileMap &tm, int spp, unsigned long long sample, int &px, int &py, int &k,
		uint32_t &slot)
{
	unsigned long long pix = sample / (unsigned)spp;
	k = (int)(sample - pix * (unsigned)spp);
	unsigned per_tile = (unsigned)(tm.tile_w * tm.tile_h);
	unsigned long long lt = pix / per_tile;
	unsigned in_tile = (unsigned)(pix - lt * per_tile);
	slot = (uint32_t)pix;
	if (lt >= (unsigned long long)tm.owned)
		return false;
	int tile = tm.tile_first + (int)lt * tm.tile_stride;
	int tx = tile % tm.tiles_x, ty = tile / tm.tiles_x;
	px = tx * tm.tile_w + (int)(in_tile % (unsigned)tm.tile_w);
	py = ty * tm.tile_h + (int)(in_tile / (unsigned)tm.tile_w);
	return px >= tm.x0 && px < tm.x1 && py >= tm.y0 && py < tm.y1;
}

// ray.c:19-44 with the jitter of ray.c:47-59 or the pixel centre of ray.c:62-65
DEV Ray camera_ray(const DevScene &sc, int i, int j, float offx, float offy)
{
	const int nx = sc.width, ny = sc.height;
	float u = (float)(sc.cam_left + sc.cam_width * (double)((float)i + offx) / (double)nx);
	float v = (float)(sc.cam_bottom + sc.cam_height * (double)((float)j + offy) / (double)ny);
	float d = sc.cam_d;
	Ray r;
	r.o = ld3(sc.cam_pos);
	r.d = vnorm(vadd(vscale((double)(-d), ld3(sc.cam_w)),
				vadd(vscale((double)u, ld3(sc.cam_u)), vscale((double)v, ld3(sc.cam_v)))));
	r.tnear = 0;
	r.tfar = __builtin_huge_valf();
	return r;
}

DEV void sample_offsets(const DevScene &sc, int rng_mode, const GlibcTable glibc_table, uint32_t seed,
		int px, int py, int k, uint32_t &key, float &offx, float &offy)
{
	uint32_t pixel = (uint32_t)(sc.width * py + px);
	key = key0_of(seed, pixel, (uint32_t)k);
	if (!sc.antialiasing) {
		offx = 0.5f;
		offy = 0.5f;
		return;
	}
	const int aa = sc.aa_samples;
	const float n = (float)aa;
	int p = k % aa, q = k / aa;
	double r0, r1;
	if (rng_mode == RTAMD_RNG_GLIBC) {
		// draw 2*((j*W+i)*aa^2+k)+{0,1} of the serial stream; the table holds the draws
		// from glibc_table.first on
		unsigned long long idx = 2ull * ((unsigned long long)pixel * (unsigned)sc.spp + (unsigned)k) - glibc_table.first;
		r0 = (double)glibc_table.draws[idx] / ((double)2147483647);
		r1 = (double)glibc_table.draws[idx + 1] / ((double)2147483647);
	} else {
		r0 = keyed_drand(key, PURPOSE_AA, 0);
		r1 = keyed_drand(key, PURPOSE_AA, 1);
	}
	offx = (float)(((double)p + r0) / (double)n);
	offy = (float)(((double)q + r1) / (double)n);
}
