// Shading of a path vertex (shading.c, lighting.c, texture.c:72-182): environment
// lookup, light loops with shadow rays, reflection / glossy children.
//
// Part of the device code of kernels.hip, which includes it inside its anonymous
// namespace (after <hip/hip_runtime.h>, scene_dev.h and the parts before it).

// ---- texture.c:72-116, 135-182 -------------------------------------------------

DEV c3 texel_at(const DevScene &sc, const DevCubeFace &f, int x, int y)
{
	const float4 v = reinterpret_cast<const float4 *>(sc.texels)[f.offset + (int64_t)f.width * y + x];
	return { v.x, v.y, v.z };
}

DEV c3 texture_texel(const DevScene &sc, int face, double u, double v)
{
	const DevCubeFace f = sc.faces[face];
	if (u == 1.0) u -= 1e-6;
	if (v == 1.0) v -= 1e-6;
	u = u - __builtin_floor(u);
	v = v - __builtin_floor(v);
	u *= f.width;
	v *= f.height;
	int x0 = (int)__builtin_floor(u);
	int y0 = (int)__builtin_floor(v);
	double ualpha = u - x0, valpha = v - y0;
	double ubeta = 1 - ualpha, vbeta = 1 - valpha;
	int x1 = (x0 == f.width - 1) ? x0 : x0 + 1;
	int y1 = (y0 == f.height - 1) ? y0 : y0 + 1;
	// out-of-range coordinates can only come from a NaN direction; keep the fetch in bounds
	x0 = min(max(x0, 0), f.width - 1);  x1 = min(max(x1, 0), f.width - 1);
	y0 = min(max(y0, 0), f.height - 1); y1 = min(max(y1, 0), f.height - 1);
	c3 caa = texel_at(sc, f, x0, y0), cab = texel_at(sc, f, x1, y0);
	c3 cba = texel_at(sc, f, x0, y1), cbb = texel_at(sc, f, x1, y1);
	return cadd(cscale((float)vbeta, cadd(cscale((float)ubeta, caa), cscale((float)ualpha, cab))),
			cscale((float)valpha, cadd(cscale((float)ubeta, cba), cscale((float)ualpha, cbb))));
}

DEV c3 cubemap_colour(const DevScene &sc, d3 d)
{
	int face = -1;
	float u = 0.0f, v = 0.0f;
	double dx = __builtin_fabs(d.x), dy = __builtin_fabs(d.y), dz = __builtin_fabs(d.z);
	if (dx >= dy && dx >= dz) {
		if (d.x >= 0) { u = (float)((1.0 + d.z / d.x) / 2.0); v = (float)((1.0 + d.y / d.x) / 2.0); face = 1; }
		else { u = (float)(1.0 - (1.0 - d.z / d.x) / 2.0); v = (float)((1.0 - d.y / d.x) / 2.0); face = 0; }
	} else if (dy >= dz && dy >= dx) {
		if (d.y >= 0) { u = (float)((1.0 + d.x / d.y) / 2.0); v = (float)(1.0 - (1.0 - d.z / d.y) / 2.0); face = 3; }
		else { u = (float)((1.0 - d.x / d.y) / 2.0); v = (float)((1.0 + d.z / d.y) / 2.0); face = 2; }
	} else if (dz >= dx && dz >= dy) {
		if (d.z >= 0) { u = (float)(1.0 - (1.0 + d.x / d.z) / 2.0); v = (float)((1.0 + d.y / d.z) / 2.0); face = 5; }
		else { u = (float)((1.0 - d.x / d.z) / 2.0); v = (float)((1.0 - d.y / d.z) / 2.0); face = 4; }
	}
	if (face < 0)
		return { 0.0f, 0.0f, 0.0f };
	return texture_texel(sc, face, (double)u, (double)v);
}

// ---- lighting.c ----------------------------------------------------------------

// lighting.c:7-19
DEV c3 diff_colour(CLight &light, CMaterial &mat, d3 light_dir, d3 normal)
{
	double dp = vdot(light_dir, normal);
	float ndotv = (float)(0 > dp ? 0.0 : dp);
	c3 light_col = cscale(light.intensity, cscale(ndotv, lc3(light.colour)));
	return cmul(lc3(mat.diffuse), light_col);
}

// lighting.c:21-34 (called with cam_dir and light_dir swapped, shading.c:52; the
// half vector is symmetric in them)
DEV c3 spec_colour(CLight &light, CMaterial &mat, d3 first, d3 second, d3 normal)
{
	d3 half = vnorm(vadd(first, second));
	double dp = vdot(half, normal);
	float base = (float)(0 > dp ? 0.0 : dp);
	float hdotn = __builtin_powf(base, (float)mat.shininess);
	c3 light_col = cscale(light.intensity, cscale(hdotn, lc3(light.colour)));
	return cmul(lc3(mat.specular), light_col);
}

// A light sample that faces away from the light (no diffuse term) and whose specular
// term underflows to zero adds exactly nothing to the sum, occluded or not
// (shading.c:48-54 adds the two terms of an unoccluded sample; x + 0 = x in float, and the
// sums start at +0).  Its shadow ray is counted but need not be walked: typically the
// rays that would start on the far side of a body and cross all of it.
#ifndef RT_SKIP_DARK
#define RT_SKIP_DARK 1
#endif
DEV bool sample_is_dark(CLight &light, CMaterial &mat, d3 cam_dir, d3 light_dir, d3 normal)
{
#if RT_SKIP_DARK
	if (vdot(light_dir, normal) > 0)
		return false;
	const c3 c = cadd(diff_colour(light, mat, light_dir, normal), spec_colour(light, mat, cam_dir, light_dir, normal));
	return c.r == 0.0f && c.g == 0.0f && c.b == 0.0f;
#else
	return false;
#endif
}

// ---- wave helpers ------------------------------------------------------------


// Adds per-lane counts into a global counter with one atomic per wave.
DEV void wave_count_add(unsigned long long *counter, unsigned value)
{
	unsigned total = value;
	for (int off = 32; off > 0; off >>= 1)
		total += __shfl_xor(total, off, 64);
	if (lane_id() == 0 && total)
		atomicAdd(counter, (unsigned long long)total);
}

DEV long long to_fixed(float c)
{
	// NaN and infinities cannot be accumulated; the quantiser maps them to 0 / 255 anyway
	if (!(c == c)) return 0;
	double v = (double)c;
	if (v > 1.0e6) v = 1.0e6;
	if (v < -1.0e6) v = -1.0e6;
	return __double2ll_rn(v * RT_FIX_SCALE);
}

DEV void accumulate(long long *accum, uint32_t slot, c3 c)
{
	unsigned long long *p = reinterpret_cast<unsigned long long *>(accum + 3ull * slot);
	long long r = to_fixed(c.r), g = to_fixed(c.g), b = to_fixed(c.b);
	if (r) atomicAdd(p + 0, (unsigned long long)r);
	if (g) atomicAdd(p + 1, (unsigned long long)g);
	if (b) atomicAdd(p + 2, (unsigned long long)b);
}

// The same for every lane of a wave at once (all 64 lanes must call; `on` says which
// ones have something to add).  Neighbouring lanes that add to the same pixel are
// summed first and the first lane of the run issues the three atomics: integer adds
// commute, so the pixel ends up with the same bits, and a wave whose lanes hold the
// samples of one pixel sends 3 atomics instead of 192 to one address.
DEV void accumulate_runs(long long *accum, uint32_t slot, c3 c, bool on)
{
	long long r = on ? to_fixed(c.r) : 0, g = on ? to_fixed(c.g) : 0, b = on ? to_fixed(c.b) : 0;
	const unsigned lane = lane_id();
	const uint32_t seg = on ? slot : 0xffffffffu;
	const uint32_t prev = __shfl_up(seg, 1, 64);
	const bool first = lane == 0 || prev != seg;
	// the lane a run starts at names the run
	const u64 firsts = __builtin_amdgcn_ballot_w64(first);
	const unsigned start = 63u - (unsigned)__builtin_clzll(firsts & (~0ull >> (63u - lane)));
	for (int off = 1; off < 64; off <<= 1) {
		const unsigned ostart = __shfl_down(start, off, 64);
		const long long orr = __shfl_down(r, off, 64), og = __shfl_down(g, off, 64), ob = __shfl_down(b, off, 64);
		if (lane + off < 64 && ostart == start) {
			r += orr; g += og; b += ob;
		}
	}
	if (on && first) {
		unsigned long long *p = reinterpret_cast<unsigned long long *>(accum + 3ull * slot);
		if (r) atomicAdd(p + 0, (unsigned long long)r);
		if (g) atomicAdd(p + 1, (unsigned long long)g);
		if (b) atomicAdd(p + 2, (unsigned long long)b);
	}
}

// What a path vertex needs besides its ray.
struct ShadeIO {
	const DevScene *sc;
	int rng_mode;
	QueueItem *queue_out;
	unsigned *queue_count;
	unsigned queue_capacity;
	DevCounters *counters;
	ShaftEntry *shaft_lists;     // this wave's leaf lists in LDS, or null
	int *wave_stack;             // transposed light loops: this wave's packet stack in LDS
	float *wave_sum;             // ... and 384 floats for the ordered sums
};

// ---- light samples -------------------------------------------------------------

// shading.c:24-37: the point of the light that sample j aims at.
DEV d3 light_sample_point(const DevScene &sc, CLight &light, bool area, int n, int li, uint32_t key, int j)
{
	const d3 lpos = ld3(light.position);
	if (!area)
		return lpos;
	int p = j % n, q = j / n;
	float alpha = (float)((double)((float)p / (float)n) +
			keyed_drand(key, PURPOSE_LIGHT0 + (uint32_t)li, 2u * (uint32_t)j));
	float beta = (float)((double)((float)q / (float)n) +
			keyed_drand(key, PURPOSE_LIGHT0 + (uint32_t)li, 2u * (uint32_t)j + 1u));
	return vadd(vadd(lpos, vscale((double)alpha, ld3(light.edge1))), vscale((double)beta, ld3(light.edge2)));
}

// shading.c:39-46
DEV Ray shadow_ray_towards(d3 position, d3 light_pos, d3 &light_dir)
{
	d3 to_light = vsub(light_pos, position);
	light_dir = vnorm(to_light);
	Ray shadow;
	shadow.d = light_dir;
	shadow.o = vadd(position, vscale(1e-4, light_dir));
	shadow.tnear = 0;
	shadow.tfar = (float)vlen(to_light);
	return shadow;
}

DEV double readlane_f64(double v, int lane)
{
	const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
	const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, lane);
	const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), lane);
	return __builtin_bit_cast(double, (unsigned long long)lo | ((unsigned long long)hi << 32));
}
DEV d3 readlane_d3(d3 v, int lane) { return { readlane_f64(v.x, lane), readlane_f64(v.y, lane), readlane_f64(v.z, lane) }; }

// The light loops of a wave whose lanes hold UNRELATED hit points (bounced rays).
// Per lane, the shadow rays of one hit walk the BVH one after the other and the
// lanes of the wave diverge.  Transposed, the wave takes one hit at a time and
// its lanes are that hit's light samples: up to 64 shadow rays that leave the
// same point towards the same light, i.e. a coherent packet (scalar node and
// triangle fetches, one traversal for all of them).  The samples' contributions
// are then added in sample order (shading.c:18-54 adds them one by one).  Which
// form is cheaper depends on samples per light (m) against hits in the wave (H):
// transposed costs H packet traversals per 64 samples, per lane costs m
// divergent ones; the form is chosen per wave and light.
#ifndef RT_TRANSPOSE_GAIN
#define RT_TRANSPOSE_GAIN 4     // a divergent per-lane step is taken to cost this many packet steps
#endif

DEV d3 shfl_d3(d3 v, int src)
{
	return { __shfl(v.x, src, 64), __shfl(v.y, src, 64), __shfl(v.z, src, 64) };
}

#ifndef RT_TRANSPOSE_MAX_GROUP
#define RT_TRANSPOSE_MAX_GROUP 8     // most hit points whose samples share one packet
#endif

template <typename Unused = void>
DEV c3 lights_transposed(const ShadeIO &io, bool lit, d3 position, d3 normal, d3 cam_dir, int material,
		uint32_t key, int *lane_stack, unsigned &n_shadow)
{
	const DevScene &sc = *io.sc;
	const unsigned lane = lane_id();
	float *sum = io.wave_sum;
	c3 total = { 0.0f, 0.0f, 0.0f };
	const u64 lit_lanes = __builtin_amdgcn_ballot_w64(lit);
	const int hits = __builtin_popcountll(lit_lanes);
	TM_BEGIN(t_all);
	for (int li = 0; li < sc.num_lights; li++) {
		CLight &light = in_const(sc.lights)[li];
		const bool area = light.type == RTAMD_LIGHT_AREA;
		const int n = area ? sc.shadow_samples : 1;
		const int m = n * n;
		const float inv = (float)(1.0 / (double)(n * n));
		// hit points per packet: as many runs of m lanes as fit into the wave
		const int group = m >= 64 ? 1 : 64 / m;
		if (group > RT_TRANSPOSE_MAX_GROUP || (m >= 64 && m * RT_TRANSPOSE_GAIN < hits)) {
			// per lane, as shade_vertex does it
			if (lit) {
				CMaterial &mat = in_const(sc.materials)[material];
				c3 light_total = { 0.0f, 0.0f, 0.0f };
				for (int j = 0; j < m; j++) {
					d3 light_dir;
					const Ray shadow = shadow_ray_towards(position, light_sample_point(sc, light, area, n, li, key, j), light_dir);
					n_shadow++;
					Best dummy;
					if (sample_is_dark(light, mat, cam_dir, light_dir, normal))
						continue;
					if (trace<true, false>(sc, shadow, lane_stack, dummy, shaft_none()))
						continue;
					light_total = cadd(light_total, cadd(diff_colour(light, mat, light_dir, normal),
							spec_colour(light, mat, cam_dir, light_dir, normal)));
				}
				total = cadd(total, cscale(inv, light_total));
			}
			continue;
		}
		if (m < 64) {
			// `group` hit points at a time: lane = (run, sample)
			const int run = (int)lane / m, j = (int)lane - run * m;
			u64 todo = lit_lanes;
			while (todo) {
				const u64 members = todo;
				int src = 0, taken = 0;
				for (int k = 0; k < group && todo; k++) {
					const int hk = __builtin_ctzll(todo);
					todo &= todo - 1;
					if (run == k)
						src = hk;
					taken++;
				}
				const bool valid = run < taken;
				// the hit point of this lane's run
				const d3 hp = shfl_d3(position, src), hn = shfl_d3(normal, src), hc = shfl_d3(cam_dir, src);
				const uint32_t hk = (uint32_t)__shfl((int)key, src, 64);
				const int hm = __shfl(material, src, 64);
				c3 c = { 0.0f, 0.0f, 0.0f };
				if (valid) {
					CMaterial &mat = in_const(sc.materials)[hm];
					d3 light_dir;
					TM_BEGIN(t_gen);
					const Ray shadow = shadow_ray_towards(hp, light_sample_point(sc, light, area, n, li, hk, j), light_dir);
					TM_END(TM_LIGHT_SETUP, t_gen);
					Best dummy;
					TM_BEGIN(t_tr);
					bool occluded = true;
					if (!sample_is_dark(light, mat, hc, light_dir, hn))
						occluded = trace<true, true>(sc, shadow, io.wave_stack, dummy, shaft_none());
					TM_END(TM_TRACE_SHADOW, t_tr);
					TM_BEGIN(t_sh);
					if (!occluded)
						c = cadd(diff_colour(light, mat, light_dir, hn), spec_colour(light, mat, hc, light_dir, hn));
					TM_END(TM_SHADE, t_sh);
				}
				TM_BEGIN(t_sum);
				// per run and channel: ((0 + c_0) + c_1) + ... in sample order (shading.c:18-54;
				// an occluded sample adds +0, which leaves the non-negative sum unchanged)
				sum[lane] = c.r; sum[64 + lane] = c.g; sum[128 + lane] = c.b;
				for (int task = (int)lane; task < 3 * taken; task += 64) {
					const int r = task / 3, ch = task - 3 * r;
					const float *src_run = sum + 64 * ch + r * m;
					float a = src_run[0];
					for (int i = 1; i < m; i++)
						a += src_run[i];
					sum[192 + task] = a;
				}
				u64 rest = members;
				for (int k = 0; k < taken; k++) {
					const int hk2 = __builtin_ctzll(rest);
					rest &= rest - 1;
					if ((int)lane == hk2) {
						total = cadd(total, cscale(inv, { sum[192 + 3 * k], sum[193 + 3 * k], sum[194 + 3 * k] }));
						n_shadow += (unsigned)m;
					}
				}
				TM_END(TM_TAIL, t_sum);
			}
			continue;
		}
		// 64 or more samples per light: one hit point at a time, 64 samples per packet
		u64 todo = lit_lanes;
		while (todo) {
			const int h = __builtin_ctzll(todo);
			todo &= todo - 1;
			const d3 hp = readlane_d3(position, h), hn = readlane_d3(normal, h), hc = readlane_d3(cam_dir, h);
			const uint32_t hk = (uint32_t)__builtin_amdgcn_readlane((int)key, h);
			CMaterial &mat = in_const(sc.materials)[__builtin_amdgcn_readlane(material, h)];
			c3 acc = { 0.0f, 0.0f, 0.0f };      // running sum, identical in every lane
			for (int base = 0; base < m; base += 64) {
				const int j = base + (int)lane;
				c3 c = { 0.0f, 0.0f, 0.0f };
				if (j < m) {
					d3 light_dir;
					const Ray shadow = shadow_ray_towards(hp, light_sample_point(sc, light, area, n, li, hk, j), light_dir);
					Best dummy;
					if (!sample_is_dark(light, mat, hc, light_dir, hn) && !trace<true, true>(sc, shadow, io.wave_stack, dummy, shaft_none()))
						c = cadd(diff_colour(light, mat, light_dir, hn), spec_colour(light, mat, hc, light_dir, hn));
				}
				const int cnt = m - base < 64 ? m - base : 64;
				sum[lane] = c.r; sum[64 + lane] = c.g; sum[128 + lane] = c.b;
				if (lane < 3) {
					float a = lane == 0 ? acc.r : lane == 1 ? acc.g : acc.b;
					for (int i = 0; i < cnt; i++)
						a += sum[64 * lane + i];
					sum[192 + lane] = a;
				}
				acc = { sum[192], sum[193], sum[194] };
			}
			if ((int)lane == h) {
				total = cadd(total, cscale(inv, acc));
				n_shadow += (unsigned)m;
			}
		}
	}
	TM_END(TM_SHAFT, t_all);
	return total;
}

// `ray_again` returns the incoming direction once more (recomputed or re-read by
// the caller): it is needed again only when a reflection is spawned, after all the
// shadow rays, and is not worth six registers until then.
//
// shade_hit is the part of a path vertex after its nearest hit is known: the light
// loops and the children.  `lit` = this lane holds a hit; every lane of the wave
// must call it (wave-wide compaction, shafts and transposed loops need all 64).
template <bool PACKET, bool TRANSPOSE, typename RayAgain>
DEV c3 shade_hit(const ShadeIO &io, bool lit, int surface, d3 position, d3 normal, d3 ray_d, uint32_t depth,
		uint32_t key, c3 weight, uint32_t slot, int *stack, unsigned &n_shadow, RayAgain ray_again)
{
	const DevScene &sc = *io.sc;
	c3 total = { 0.0f, 0.0f, 0.0f };
	// bounds of the wave's hit points (origins of its shadow rays), rounded outwards
	float olo[3] = { 0, 0, 0 }, ohi[3] = { 0, 0, 0 };
	const bool shafts = RT_SHAFT && PACKET && io.shaft_lists && sc.any_area_light;
	TM_BEGIN(t_bounds);
	if (shafts) {
		const float inf = __builtin_huge_valf();
		const double pv[3] = { position.x, position.y, position.z };
		for (int a = 0; a < 3; a++) {
			const float f = (float)pv[a];
			const float w = __builtin_fabsf(f) * 1.1920929e-7f + 1e-37f;
			float lo = lit ? f - w : inf, hi = lit ? f + w : -inf;
			for (int off = 32; off > 0; off >>= 1) {
				lo = __builtin_fminf(lo, __shfl_xor(lo, off, 64));
				hi = __builtin_fmaxf(hi, __shfl_xor(hi, off, 64));
			}
			olo[a] = as_f(__builtin_amdgcn_readfirstlane(as_i(lo)));
			ohi[a] = as_f(__builtin_amdgcn_readfirstlane(as_i(hi)));
		}
	}
	TM_END(TM_BOUNDS, t_bounds);
#ifdef RT_TRAVERSAL_STATS
	if (shafts && depth > 0 && ohi[0] >= olo[0]) {
		const float ext = __builtin_fmaxf(__builtin_fmaxf(ohi[0] - olo[0], ohi[1] - olo[1]), ohi[2] - olo[2]);
		int k = 0;
		while (k < 15 && ext > 0.001f * (float)(2 << k))
			k++;
		RT_EV(EV_BOX_SIZE0 + k);
	}
#endif

	int n_children = 0;
	CMaterial *mat = nullptr;
	d3 cam_dir = { 0, 0, 0 };
	int material = 0;
	if (lit) {
		material = in_const(sc.surfaces)[surface].material;
		mat = &in_const(sc.materials)[material];
		cam_dir = vnorm(vscale(-1, ray_d));                          // shading.c:118
		// shading.c:133-135 -> hit_light_colour, shading.c:7-58
		for (int li = 0; !TRANSPOSE && li < sc.num_lights; li++) {
			CLight &light = in_const(sc.lights)[li];
			const bool area = light.type == RTAMD_LIGHT_AREA;
			const int n = area ? sc.shadow_samples : 1;
			c3 light_total = { 0.0f, 0.0f, 0.0f };
			ShaftState shaft_state = shaft_none();
			const int m = n * n;
			if (shafts && area && n >= 2) {
				TM_BEGIN(t_shaft);
#ifdef RT_TRAVERSAL_STATS
				const u64 lit_mask = __builtin_amdgcn_ballot_w64(lit);
				const int first_surface = __builtin_amdgcn_readlane(surface, __builtin_ctzll(lit_mask | (1ull << 63)));
				const int stat_tag = (__builtin_amdgcn_ballot_w64(lit && surface != first_surface) != 0 ? 1 : 0) + (depth > 0 ? 2 : 0)
					+ (__builtin_popcountll(lit_mask) < 32 ? 4 : 0);
				shaft_state = shaft_build(sc, light, n, olo, ohi, stack, io.shaft_lists, stat_tag);
#else
				shaft_state = shaft_build(sc, light, n, olo, ohi, stack, io.shaft_lists);
#endif
				TM_END(TM_SHAFT, t_shaft);
			}
			for (int j = 0; j < m; j++) {
				TM_BEGIN(t_lsetup);
				d3 light_dir;
				const Ray shadow = shadow_ray_towards(position, light_sample_point(sc, light, area, n, li, key, j), light_dir);
				n_shadow++;
				RT_EV(EV_SHADOW_RAYS);
				Best dummy;
				const bool dark = sample_is_dark(light, *mat, cam_dir, light_dir, normal);
				TM_END(TM_LIGHT_SETUP, t_lsetup);
#ifdef RT_TRAVERSAL_STATS
				if (__builtin_amdgcn_ballot_w64(!dark) == 0)
					RT_EV(EV_DARK_PACKETS);
#endif
				if (dark) {
					RT_EV_ADD(EV_DARK, __builtin_popcountll(__builtin_amdgcn_ballot_w64(true)));
					continue;
				}
				TM_BEGIN(t_shadow);
				const bool occluded = trace<true, PACKET>(sc, shadow, stack, dummy, shaft_state);
				TM_END(TM_TRACE_SHADOW, t_shadow);
				if (occluded)
					continue;
				TM_BEGIN(t_shade);
				RT_EV(EV_SHADE);
				c3 diff_col = diff_colour(light, *mat, light_dir, normal);
				c3 spec_col = spec_colour(light, *mat, cam_dir, light_dir, normal);
				light_total = cadd(light_total, cadd(diff_col, spec_col));
				TM_END(TM_SHADE, t_shade);
			}
			light_total = cscale((float)(1.0 / (double)(n * n)), light_total);
			total = cadd(total, light_total);
		}
		// shading.c:70-112: children exist when the material reflects and the child's
		// depth passes the cut at shading.c:120
		if (mat->reflect > 0.0f && (int)depth + 1 <= sc.max_reflections)
			n_children = (mat->glossiness <= 0.0f || depth > 1) ? 1 : sc.reflection_samples;
	}
	if (TRANSPOSE) {
		// every lane of the wave takes part, whether it holds a hit or not
		const c3 lit_total = lights_transposed(io, lit, position, normal, cam_dir, material, key, stack, n_shadow);
		if (lit)
			total = lit_total;
	}

	// Compact the spawning lanes: one slot reservation per wave.
	TM_BEGIN(t_spawn);
	if (io.queue_out) {
		unsigned want = (unsigned)n_children;
		unsigned prefix = want;
		const unsigned lane = lane_id();
		// inclusive prefix sum over the wave
		for (int off = 1; off < 64; off <<= 1) {
			unsigned v = __shfl_up(prefix, off, 64);
			if (lane >= (unsigned)off)
				prefix += v;
		}
		unsigned wave_total = __shfl(prefix, 63, 64);
		if (wave_total) {
			unsigned base = 0;
			if (lane == 0)
				base = atomicAdd(io.queue_count, wave_total);
			base = __shfl(base, 0, 64);
			unsigned my = base + prefix - want;
			if (want) {
				if (my + want > io.queue_capacity) {
					// The children are dropped and the frame is reported incomplete.  The
					// consumer reads min(count, capacity) entries: those of this lane's
					// reserved slots that lie inside the queue are marked unused.
					atomicAdd(&io.counters->queue_overflow, 1ull);
					for (unsigned i = my; i < my + want && i < io.queue_capacity; i++)
						io.queue_out[i].depth = QUEUE_SLOT_UNUSED;
				} else {
					// shading.c:81-84
					const d3 in_dir = ray_again();
					double dn = vdot(in_dir, normal);
					d3 rdir = vadd(in_dir, vscale(-2 * dn, normal));          // vector.c:77-81
					d3 rorg = vadd(position, vscale(1e-2, rdir));
					c3 w = cmul(lc3(mat->specular), { mat->reflect, mat->reflect, mat->reflect });
					w = cmul(w, weight);
					if (want == 1 && (mat->glossiness <= 0.0f || depth > 1)) {
						QueueItem it;
						it.o[0] = rorg.x; it.o[1] = rorg.y; it.o[2] = rorg.z;
						it.d[0] = rdir.x; it.d[1] = rdir.y; it.d[2] = rdir.z;
						it.weight[0] = w.r; it.weight[1] = w.g; it.weight[2] = w.b;
						it.slot = slot; it.key = key_child(key, 0); it.depth = depth + 1;
						it.pad[0] = it.pad[1] = 0;
						io.queue_out[my] = it;
					} else {
						// shading.c:93-109
						const float inv = (float)(1. / (double)sc.reflection_samples);
						w = cscale(inv, w);
						d3 pdir = vnorm(rdir);
						const d3 n1 = { 1, 0, 0 }, n2 = { 0, 1, 0 };
						d3 orth = (__builtin_fabs(vdot(pdir, n1)) < vlen(pdir) / 1.4142135623730950488016887242096980785696718753769480731766)
								? vcross(pdir, n1) : vcross(pdir, n2);            // shading.c:60-68
						d3 ta = vnorm(orth);
						d3 tb = vnorm(vcross(pdir, ta));
						for (unsigned i = 0; i < want; i++) {
							double ra = keyed_drand(key, PURPOSE_GLOSSY, 2u * i);
							double rb = keyed_drand(key, PURPOSE_GLOSSY, 2u * i + 1u);
							d3 a = vscale((double)mat->glossiness * (2 * ra - 1), ta);
							d3 b = vscale((double)mat->glossiness * (2 * rb - 1), tb);
							d3 dir = vadd(pdir, vadd(a, b));
							QueueItem it;
							it.o[0] = rorg.x; it.o[1] = rorg.y; it.o[2] = rorg.z;
							it.d[0] = dir.x; it.d[1] = dir.y; it.d[2] = dir.z;
							it.weight[0] = w.r; it.weight[1] = w.g; it.weight[2] = w.b;
							it.slot = slot; it.key = key_child(key, i); it.depth = depth + 1;
							it.pad[0] = it.pad[1] = 0;
							io.queue_out[my + i] = it;
						}
					}
				}
			}
		}
	}
	TM_END(TM_SPAWN, t_spawn);
	return total;
}

// ---- shading of one path vertex: shading.c:114-141 without the recursion ------
//
// Returns the radiance this vertex contributes directly (environment, or the sum
// of the light loops) and pushes reflection / glossy children into the bounce
// queue with their throughput weight.
template <bool PACKET, bool TRANSPOSE, typename RayAgain>
DEV c3 shade_vertex(const ShadeIO &io, const Ray &ray, uint32_t depth, uint32_t key, c3 weight,
		uint32_t slot, bool active, int *stack, rtamd_hit_record *rec,
		unsigned &n_shadow, unsigned &n_env, RayAgain ray_again)
{
	const DevScene &sc = *io.sc;
	Best best;
	bool hit = false;
	d3 position = { 0, 0, 0 }, normal = { 0, 0, 0 };
	if (active) {
		TM_BEGIN(t_trace);
		hit = trace<false, PACKET>(sc, ray, stack, best, shaft_none());
		TM_END(TM_TRACE_PRIMARY, t_trace);
		TM_BEGIN(t_finish);
		if (hit)
			finish_hit(sc, ray, best, position, normal);
		TM_END(TM_FINISH, t_finish);
	}
	if (rec && active) {
		rtamd_hit_record r;
		r.surface = hit ? best.surface : -1;
		r.triangle = -1;
		if (hit && best.ncode == NC_MESH) {
			CMesh &mesh = in_const(sc.meshes)[in_const(sc.surfaces)[best.surface].mesh];
			r.triangle = as_i(reinterpret_cast<const float4 *>(sc.tris + mesh.tri_first)[3 * best.tri].w);
		}
		r.t = hit ? best.t : 0.0f;
		r.pad = 0.0f;
		r.position[0] = position.x; r.position[1] = position.y; r.position[2] = position.z;
		r.normal[0] = normal.x; r.normal[1] = normal.y; r.normal[2] = normal.z;
		*rec = r;
	}
	c3 miss_total = { 0.0f, 0.0f, 0.0f };
	if (active && !hit) {
		// shading.c:123-129
		if (sc.has_cubemap) {
			TM_BEGIN(t_env);
			miss_total = cubemap_colour(sc, ray.d);
			TM_END(TM_ENV, t_env);
			n_env++;
		} else
			miss_total = lc3(sc.background);
	}
	const c3 lit_total = shade_hit<PACKET, TRANSPOSE>(io, active && hit, best.surface, position, normal, ray.d, depth,
			key, weight, slot, stack, n_shadow, ray_again);
	return (active && !hit) ? miss_total : lit_total;
}
