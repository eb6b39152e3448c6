// HIP kernels of the render path for gfx950 (wave64).
//
// Arithmetic contract: every quantity that decides WHICH surface or triangle a
// ray hits, and the hit distance itself, is computed with the reference's own
// mixture of double vectors and float scalars, one IEEE rounding per
// operation (this file is compiled with -ffp-contract=off; the only fused
// operations are the explicitly written __builtin_fmaf calls in the BVH box
// test, which is this implementation's own conservative structure and never
// produces a reported value).  Reference lines are cited per function.
//
// Execution model: one lane per primary sample.  A wave holds 64 consecutive
// samples, i.e. for 64 samples per pixel all lanes of a wave share a pixel and
// its rays are coherent in the BVH.  Shadow rays are traced in the lane that
// owns the hit (they never leave registers); reflection / glossy rays go to a
// bounce queue in HBM: lanes that spawn rays are compacted with a wave-wide
// ballot + prefix count and the wave reserves its slots with one atomic.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "scene_dev.h"
#include "kernels.h"

namespace {


#include "rt_base.h"
#include "rt_intersect.h"
#include "rt_traverse.h"
#include "rt_shaft.h"
#include "rt_trace.h"
#include "rt_shade.h"
#include "rt_camera.h"


} // namespace

// =============================================================================
// Kernels
// =============================================================================

// Primary samples: camera ray, nearest hit, direct lighting with shadow rays,
// environment lookup; children go to the bounce queue.  raytracer.c:32-54,78-84.
// Register budget: the primary kernel is fastest at five waves per SIMD (<= 96
// VGPRs, some spilling; measured 78.9 ms vs 83.9 ms at four and 81.5 ms at six
// on the 1080p showcase frame); the bounce kernel keeps per-lane stacks in LDS
// (32 KB per workgroup) and runs at four.
#ifndef RT_MIN_WAVES
#define RT_MIN_WAVES 4
#endif
#ifndef RT_MIN_WAVES_BOUNCE
#define RT_MIN_WAVES_BOUNCE 4
#endif
#ifndef RT_BOUNCE_PACKET
#define RT_BOUNCE_PACKET 0
#endif
#ifndef RT_BOUNCE_TRANSPOSE
#define RT_BOUNCE_TRANSPOSE 1
#endif
__global__ void __launch_bounds__(RT_BLOCK, RT_MIN_WAVES)
k_primary(DevScene sc, TileMap tm, unsigned long long sample_first, unsigned long long sample_count,
		int rng_mode, uint32_t seed, const int32_t *glibc_table, long long *accum, float *direct_out,
		rtamd_hit_record *hits, QueueItem *queue_out, unsigned *queue_count, unsigned queue_capacity,
		DevCounters *counters)
{
	// one traversal stack per wave (packet traversal)
	__shared__ int4 s_stack[(RT_BLOCK / 64) * RT_MAX_STACK];
	__shared__ __attribute__((aligned(16))) float s_sum[RT_BLOCK / 64][3 * 64];
	const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // scalar register
	int *stack = reinterpret_cast<int *>(s_stack + wave * RT_MAX_STACK);
#ifdef RT_TIMING
	if ((threadIdx.x & 63) < TM_COUNT)
		s_timing[threadIdx.x >> 6][threadIdx.x & 63] = 0;
#endif
	TM_BEGIN(t_kernel);
	TM_BEGIN(t_setup);
	unsigned long long local = (unsigned long long)blockIdx.x * RT_BLOCK + threadIdx.x;
	unsigned long long sample = sample_first + local;
	int px = 0, py = 0, k = 0;
	uint32_t slot = 0;
	bool active = local < sample_count && sample_to_pixel(tm, sc.spp, sample, px, py, k, slot);

	uint32_t key = 0;
	Ray ray;
	ray.o = { 0, 0, 0 }; ray.d = { 0, 0, 1 }; ray.tnear = 0; ray.tfar = 0;
	if (active) {
		float offx, offy;
		sample_offsets(sc, rng_mode, glibc_table, seed, px, py, k, key, offx, offy);
		ray = camera_ray(sc, px, py, offx, offy);
	}
	__shared__ ShaftEntry s_shaft[RT_BLOCK / 64][RT_SHAFT_LISTS * RT_SHAFT_CAP];
	ShadeIO io = { &sc, rng_mode, queue_out, queue_count, queue_capacity, counters, s_shaft[wave], nullptr, nullptr };
	rtamd_hit_record *rec = nullptr;
	if (hits && active) {
		// window-local record index
		unsigned long long wpix = (unsigned long long)(py - tm.y0) * (unsigned)(tm.x1 - tm.x0) + (unsigned)(px - tm.x0);
		rec = hits + wpix * (unsigned)sc.spp + (unsigned)k;
	}
	unsigned n_shadow = 0, n_env = 0;
	c3 weight = { 1.0f, 1.0f, 1.0f };
	TM_END(TM_SETUP, t_setup);
	auto ray_again = [&]() {
		// same arithmetic from the same inputs: the same bits as `ray.d`
		unsigned tid = threadIdx.x;
		asm volatile("" : "+v"(tid));
		int qx = 0, qy = 0, qk = 0;
		uint32_t qslot, qkey;
		float ox, oy;
		sample_to_pixel(tm, sc.spp, sample_first + (unsigned long long)blockIdx.x * RT_BLOCK + tid, qx, qy, qk, qslot);
		sample_offsets(sc, rng_mode, glibc_table, seed, qx, qy, qk, qkey, ox, oy);
		return camera_ray(sc, qx, qy, ox, oy).d;
	};
	c3 total = shade_vertex<true, false>(io, ray, 0, key, weight, slot, active, stack, rec, n_shadow, n_env, ray_again);
	// the pixel mapping again, rather than five registers kept through the kernel
	int tx = 0, ty = 0, tk = 0;
	uint32_t tslot = 0;
	bool tactive;
	{
		unsigned tid = threadIdx.x;
		asm volatile("" : "+v"(tid));
		const unsigned long long l2 = (unsigned long long)blockIdx.x * RT_BLOCK + tid;
		tactive = l2 < sample_count && sample_to_pixel(tm, sc.spp, sample_first + l2, tx, ty, tk, tslot);
	}

	TM_BEGIN(t_tail);
	const unsigned lane = lane_id();
	if (64 % sc.spp == 0) {
		// All samples of a pixel sit in consecutive lanes of this wave: add them in
		// the reference's order, ((c0 + c1) + c2) + ... in float (raytracer.c:41-45).
		const unsigned spp = (unsigned)sc.spp;
		const unsigned base = lane - lane % spp;
		float ar = total.r, ag = total.g, ab = total.b;
		if (spp % 4 == 0) {
			// through LDS: one lane per (pixel, channel) walks its run of samples
			float *sw = s_sum[wave];
			sw[lane] = total.r; sw[64 + lane] = total.g; sw[128 + lane] = total.b;
			const unsigned ppw = 64 / spp;
			if (lane < 3 * ppw) {
				float *run = sw + 64 * (lane / ppw) + (lane % ppw) * spp;
				const float4 *src = reinterpret_cast<const float4 *>(run);
				float4 v = src[0];
				float acc = v.x;
				acc += v.y; acc += v.z; acc += v.w;
				for (unsigned j = 1; j < spp / 4; j++) {
					v = src[j];
					acc += v.x; acc += v.y; acc += v.z; acc += v.w;
				}
				run[0] = acc;
			}
			ar = sw[base]; ag = sw[64 + base]; ab = sw[128 + base];
		} else {
			for (unsigned j = 1; j < spp; j++) {
				ar += __shfl(total.r, (int)(base + j), 64);
				ag += __shfl(total.g, (int)(base + j), 64);
				ab += __shfl(total.b, (int)(base + j), 64);
			}
		}
		if (tactive && lane == base) {
			if (direct_out) {
				// no other contribution can arrive: write the pixel (raytracer.c:46,81)
				if (sc.antialiasing) {
					const float s = (float)(1.0 / (double)sc.spp);
					ar = s * ar; ag = s * ag; ab = s * ab;
				}
				float4 out = { ar, ag, ab, 0.0f };
				reinterpret_cast<float4 *>(direct_out)[(unsigned long long)sc.width * ty + tx] = out;
			} else {
				accumulate(accum, tslot, { ar, ag, ab });
			}
		}
	} else {
		// Per-pixel sum in 2^-32 fixed point: integer adds commute, so the result does
		// not depend on the order in which lanes, waves or later bounces arrive.
		long long r = tactive ? to_fixed(total.r) : 0, g = tactive ? to_fixed(total.g) : 0,
		     b = tactive ? to_fixed(total.b) : 0;
		// segmented reduction over runs of lanes that share a pixel (runs are contiguous)
		uint32_t seg = tactive ? tslot : 0xffffffffu;
		for (int off = 1; off < 64; off <<= 1) {
			uint32_t oseg = __shfl_down(seg, off, 64);
			long long orr = __shfl_down(r, off, 64), og = __shfl_down(g, off, 64), ob = __shfl_down(b, off, 64);
			// lane + off belongs to my run only if every lane in between does, which holds
			// for contiguous runs when its slot equals mine
			if (lane + off < 64 && oseg == seg) {
				r += orr; g += og; b += ob;
			}
		}
		uint32_t prev = __shfl_up(seg, 1, 64);
		bool head = tactive && (lane == 0 || prev != seg);
		if (head) {
			unsigned long long *p = reinterpret_cast<unsigned long long *>(accum + 3ull * tslot);
			if (r) atomicAdd(p + 0, (unsigned long long)r);
			if (g) atomicAdd(p + 1, (unsigned long long)g);
			if (b) atomicAdd(p + 2, (unsigned long long)b);
		}
	}
	const u64 active_lanes = __builtin_amdgcn_ballot_w64(tactive), env_lanes = __builtin_amdgcn_ballot_w64(n_env != 0);
	if (lane == 0) {
		if (active_lanes) atomicAdd(&counters->rays_primary, (unsigned long long)__builtin_popcountll(active_lanes));
		if (env_lanes) atomicAdd(&counters->env_lookups, (unsigned long long)__builtin_popcountll(env_lanes));
	}
	wave_count_add(&counters->rays_shadow, n_shadow);
	TM_END(TM_TAIL, t_tail);
	TM_END(TM_KERNEL, t_kernel);
#ifdef RT_TIMING
	if ((threadIdx.x & 63) < TM_COUNT)
		atomicAdd(&g_timing[threadIdx.x & 63], s_timing[threadIdx.x >> 6][threadIdx.x & 63]);
#endif
}

// One bounce: every queued ray is traced and shaded like a primary one
// (shading.c:114-141 at depth >= 1) and its radiance, scaled by the path
// weight, is added to its pixel.  Grid-stride over the queue population, which
// is read from device memory so the host never waits between bounces.
__global__ void __launch_bounds__(RT_BLOCK, RT_MIN_WAVES_BOUNCE)
k_bounce(DevScene sc, const QueueItem *queue_in, const unsigned *queue_in_count, unsigned queue_capacity,
		long long *accum, QueueItem *queue_out, unsigned *queue_out_count, DevCounters *counters)
{
#if RT_BOUNCE_PACKET
	// consecutive queue entries come from the lanes of one spawning wave (one
	// pixel's samples): they are coherent enough to walk the BVH as a packet
	__shared__ int4 s_stack[(RT_BLOCK / 64) * RT_MAX_STACK];
	__shared__ ShaftEntry s_shaft[RT_BLOCK / 64][RT_SHAFT_LISTS * RT_SHAFT_CAP];
	int *stack = reinterpret_cast<int *>(s_stack + (threadIdx.x >> 6) * RT_MAX_STACK);
	ShaftEntry *shaft_lists = s_shaft[threadIdx.x >> 6];
#else
	__shared__ int s_stack[RT_MAX_STACK * RT_BLOCK];
	int *stack = s_stack + threadIdx.x;
	ShaftEntry *shaft_lists = nullptr;
#endif
#ifdef RT_TIMING
	if ((threadIdx.x & 63) < TM_COUNT)
		s_timing[threadIdx.x >> 6][threadIdx.x & 63] = 0;
#endif
	TM_BEGIN(t_kernel);
	unsigned count = *queue_in_count;
	if (count > queue_capacity)
		count = queue_capacity;
	if (blockIdx.x == 0 && threadIdx.x == 0)
		atomicMax(&counters->queue_peak, (unsigned long long)count);
	const unsigned stride = gridDim.x * RT_BLOCK;
	// every lane of a wave runs the same number of iterations (the wave-level
	// compaction inside shade_vertex needs all 64 lanes present)
	const unsigned rounds = (count + stride - 1) / stride;
	__shared__ int4 s_wave_stack[(RT_BLOCK / 64) * RT_MAX_STACK];
	__shared__ float s_wave_sum[RT_BLOCK / 64][384];
	ShadeIO io = { &sc, RTAMD_RNG_KEYED, queue_out, queue_out_count, queue_capacity, counters, shaft_lists,
		reinterpret_cast<int *>(s_wave_stack + (threadIdx.x >> 6) * RT_MAX_STACK), s_wave_sum[threadIdx.x >> 6] };
	unsigned n_shadow = 0, n_env = 0, n_rays = 0;
	for (unsigned round = 0; round < rounds; round++) {
		unsigned idx = round * stride + blockIdx.x * RT_BLOCK + threadIdx.x;
		bool active = idx < count;
		Ray ray;
		ray.o = { 0, 0, 0 }; ray.d = { 0, 0, 1 }; ray.tnear = 0; ray.tfar = 0;
		c3 weight = { 0, 0, 0 };
		uint32_t slot = 0, key = 0, depth = 0;
		if (active) {
			const QueueItem it = queue_in[idx];
			ray.o = { it.o[0], it.o[1], it.o[2] };
			ray.d = { it.d[0], it.d[1], it.d[2] };
			ray.tnear = 0;
			ray.tfar = __builtin_huge_valf();
			weight = { it.weight[0], it.weight[1], it.weight[2] };
			slot = it.slot; key = it.key; depth = it.depth;
			n_rays++;
		}
		auto ray_again = [&]() {
			unsigned i = idx;
			asm volatile("" : "+v"(i));
			const double *dq = queue_in[i].d;
			return d3{ dq[0], dq[1], dq[2] };
		};
		c3 total = shade_vertex<RT_BOUNCE_PACKET != 0, RT_BOUNCE_TRANSPOSE != 0>(io, ray, depth, key, weight, slot, active, stack, nullptr, n_shadow, n_env, ray_again);
		if (active)
			accumulate(accum, slot, cmul(weight, total));
	}
	wave_count_add(&counters->rays_secondary, n_rays);
	wave_count_add(&counters->rays_shadow, n_shadow);
	wave_count_add(&counters->env_lookups, n_env);
	TM_END(TM_KERNEL, t_kernel);
#ifdef RT_TIMING
	if ((threadIdx.x & 63) < TM_COUNT)
		atomicAdd(&g_timing_bounce[threadIdx.x & 63], s_timing[threadIdx.x >> 6][threadIdx.x & 63]);
#endif
}

#if defined(RT_TRAVERSAL_STATS) || defined(RT_TIMING)
__global__ void k_collect_stats(DevCounters *counters)
{
#ifdef RT_TRAVERSAL_STATS
	counters->bvh_nodes = g_stat_nodes;
	counters->tri_tests = g_stat_tris;
	g_stat_nodes = 0;
	g_stat_tris = 0;
	for (int i = 0; i < RT_SHAFT_CAP + 8; i++) {
		if (g_stat_shaft[i])
			printf("shaft %d %llu\n", i, g_stat_shaft[i]);
		g_stat_shaft[i] = 0;
	}
	for (int i = 0; i < 16; i++) {
		if (g_stat_ev[i])
			printf("event %d %llu\n", i, g_stat_ev[i]);
		g_stat_ev[i] = 0;
	}
#endif
#ifdef RT_TIMING
	for (int i = 0; i < TM_COUNT; i++) {
		printf("timing %d %llu bounce %llu\n", i, g_timing[i], g_timing_bounce[i]);
		g_timing[i] = 0;
		g_timing_bounce[i] = 0;
	}
#endif
}
#endif

// Accumulators -> the reference's float RGBA frame (raytracer.c:46,81): the
// per-pixel sum is scaled by the float 1.0/aa^2 when antialiasing is on.
__global__ void k_resolve(DevScene sc, TileMap tm, unsigned long long pixel_first, unsigned long long pixel_count,
		const long long *accum, float *colour)
{
	unsigned long long local = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (local >= pixel_count)
		return;
	unsigned long long pix = pixel_first + local;
	int px, py, k;
	uint32_t slot;
	if (!sample_to_pixel(tm, 1, pix, px, py, k, slot))
		return;
	const long long *a = accum + 3ull * slot;
	float r = (float)((double)a[0] / RT_FIX_SCALE);
	float g = (float)((double)a[1] / RT_FIX_SCALE);
	float b = (float)((double)a[2] / RT_FIX_SCALE);
	if (sc.antialiasing) {
		float s = (float)(1.0 / (double)sc.spp);
		r = s * r; g = s * g; b = s * b;
	}
	float4 out = { r, g, b, 0.0f };
	reinterpret_cast<float4 *>(colour)[(unsigned long long)sc.width * py + px] = out;
}

// ray_intersect on a batch of caller-supplied rays (ray.h:25, ray.c:643-672).
__global__ void __launch_bounds__(RT_BLOCK)
k_trace(DevScene sc, const double *rays, unsigned long long n, rtamd_hit_record *hits)
{
	__shared__ int s_stack[RT_MAX_STACK * RT_BLOCK];
	int *stack = s_stack + threadIdx.x;
	unsigned long long i = (unsigned long long)blockIdx.x * RT_BLOCK + threadIdx.x;
	if (i >= n)
		return;
	const double *p = rays + 8 * i;
	Ray ray;
	ray.o = { p[0], p[1], p[2] };
	ray.d = { p[3], p[4], p[5] };
	ray.tnear = (float)p[6];
	ray.tfar = (float)p[7];
	Best best;
	rtamd_hit_record r;
	r.surface = -1; r.triangle = -1; r.t = 0; r.pad = 0;
	r.position[0] = r.position[1] = r.position[2] = 0;
	r.normal[0] = r.normal[1] = r.normal[2] = 0;
	if (trace<false, false>(sc, ray, stack, best, shaft_none())) {
		d3 position, normal;
		finish_hit(sc, ray, best, position, normal);
		r.surface = best.surface;
		if (best.ncode == NC_MESH) {
			CMesh &mesh = in_const(sc.meshes)[in_const(sc.surfaces)[best.surface].mesh];
			r.triangle = as_i(reinterpret_cast<const float4 *>(sc.tris + mesh.tri_first)[3 * best.tri].w);
		}
		r.t = best.t;
		r.position[0] = position.x; r.position[1] = position.y; r.position[2] = position.z;
		r.normal[0] = normal.x; r.normal[1] = normal.y; r.normal[2] = normal.z;
	}
	hits[i] = r;
}

// camera_ray / camera_ray_aa for every sample of the selection (ray.h:23-24).
__global__ void k_camera_rays(DevScene sc, TileMap tm, unsigned long long sample_count, int rng_mode,
		uint32_t seed, const int32_t *glibc_table, double *rays)
{
	unsigned long long sample = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (sample >= sample_count)
		return;
	int px, py, k;
	uint32_t slot;
	if (!sample_to_pixel(tm, sc.spp, sample, px, py, k, slot))
		return;
	uint32_t key;
	float offx, offy;
	sample_offsets(sc, rng_mode, glibc_table, seed, px, py, k, key, offx, offy);
	Ray ray = camera_ray(sc, px, py, offx, offy);
	double *o = rays + 8 * (((unsigned long long)sc.width * py + px) * (unsigned)sc.spp + (unsigned)k);
	o[0] = ray.o.x; o[1] = ray.o.y; o[2] = ray.o.z;
	o[3] = ray.d.x; o[4] = ray.d.y; o[5] = ray.d.z;
	o[6] = (double)ray.tnear; o[7] = (double)ray.tfar;
}

// Frame <-> dense per-owner tile buffer (multi-GPU framebuffer exchange).
__global__ void k_tiles_copy(int width, int height, TileMap tm, float *frame, float *packed, int unpack)
{
	unsigned long long pix = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
	unsigned per_tile = (unsigned)(tm.tile_w * tm.tile_h);
	if (pix >= (unsigned long long)tm.owned * per_tile)
		return;
	int px, py, k;
	uint32_t slot;
	TileMap whole = tm;
	whole.x0 = 0; whole.y0 = 0; whole.x1 = width; whole.y1 = height;
	if (!sample_to_pixel(whole, 1, pix, px, py, k, slot))
		return;
	float4 *f = reinterpret_cast<float4 *>(frame) + (unsigned long long)width * py + px;
	float4 *p = reinterpret_cast<float4 *>(packed) + pix;
	if (unpack)
		*f = *p;
	else
		*p = *f;
}

// =============================================================================
// Host-side launchers (called from api.hip)
// =============================================================================

static TileMap make_tile_map(int width, int height, const rtamd_render_params &p)
{
	TileMap tm;
	tm.tile_w = p.tile_w;
	tm.tile_h = p.tile_h;
	tm.tiles_x = (width + p.tile_w - 1) / p.tile_w;
	tm.tiles_y = (height + p.tile_h - 1) / p.tile_h;
	tm.tile_first = p.tile_first;
	tm.tile_stride = p.tile_stride;
	int total = tm.tiles_x * tm.tiles_y;
	tm.owned = p.tile_first < total ? (total - p.tile_first + p.tile_stride - 1) / p.tile_stride : 0;
	bool whole = p.x0 == 0 && p.y0 == 0 && p.x1 == 0 && p.y1 == 0;
	tm.x0 = whole ? 0 : p.x0;
	tm.y0 = whole ? 0 : p.y0;
	tm.x1 = whole ? width : p.x1;
	tm.y1 = whole ? height : p.y1;
	return tm;
}

int rt_tiles_owned(int width, int height, const rtamd_render_params &p)
{
	return make_tile_map(width, height, p).owned;
}

unsigned long long rt_owned_pixels(int width, int height, const rtamd_render_params &p)
{
	TileMap tm = make_tile_map(width, height, p);
	return (unsigned long long)tm.owned * (unsigned)(tm.tile_w * tm.tile_h);
}

hipError_t rt_launch_primary(const DevScene &sc, const rtamd_render_params &p,
		unsigned long long sample_first, unsigned long long sample_count, const int32_t *glibc_table,
		long long *accum, float *direct_out, rtamd_hit_record *hits, QueueItem *queue_out,
		unsigned *queue_count, unsigned queue_capacity, DevCounters *counters, hipStream_t stream)
{
	if (!sample_count)
		return hipSuccess;
	TileMap tm = make_tile_map(sc.width, sc.height, p);
	unsigned long long blocks = (sample_count + RT_BLOCK - 1) / RT_BLOCK;
	hipLaunchKernelGGL(k_primary, dim3((unsigned)blocks), dim3(RT_BLOCK), 0, stream, sc, tm, sample_first,
			sample_count, p.rng_mode, p.seed, glibc_table, accum, direct_out, hits, queue_out, queue_count,
			queue_capacity, counters);
	return hipGetLastError();
}

hipError_t rt_launch_bounce(const DevScene &sc, const QueueItem *queue_in, const unsigned *queue_in_count,
		unsigned queue_capacity, long long *accum, QueueItem *queue_out, unsigned *queue_out_count,
		DevCounters *counters, unsigned grid_blocks, hipStream_t stream)
{
	hipLaunchKernelGGL(k_bounce, dim3(grid_blocks), dim3(RT_BLOCK), 0, stream, sc, queue_in, queue_in_count,
			queue_capacity, accum, queue_out, queue_out_count, counters);
	return hipGetLastError();
}

hipError_t rt_launch_resolve(const DevScene &sc, const rtamd_render_params &p, unsigned long long pixel_first,
		unsigned long long pixel_count, const long long *accum, float *colour, DevCounters *counters,
		hipStream_t stream)
{
	if (!pixel_count)
		return hipSuccess;
	TileMap tm = make_tile_map(sc.width, sc.height, p);
	unsigned long long blocks = (pixel_count + 255) / 256;
	hipLaunchKernelGGL(k_resolve, dim3((unsigned)blocks), dim3(256), 0, stream, sc, tm, pixel_first, pixel_count,
			accum, colour);
	return hipGetLastError();
}

// Build options only: moves the traversal statistics into the counter block.
hipError_t rt_launch_collect_stats(DevCounters *counters, hipStream_t stream)
{
#if defined(RT_TRAVERSAL_STATS) || defined(RT_TIMING)
	hipLaunchKernelGGL(k_collect_stats, dim3(1), dim3(1), 0, stream, counters);
	return hipGetLastError();
#else
	(void)counters; (void)stream;
	return hipSuccess;
#endif
}

hipError_t rt_launch_trace(const DevScene &sc, const double *rays, unsigned long long n,
		rtamd_hit_record *hits, hipStream_t stream)
{
	if (!n)
		return hipSuccess;
	unsigned long long blocks = (n + RT_BLOCK - 1) / RT_BLOCK;
	hipLaunchKernelGGL(k_trace, dim3((unsigned)blocks), dim3(RT_BLOCK), 0, stream, sc, rays, n, hits);
	return hipGetLastError();
}

hipError_t rt_launch_camera_rays(const DevScene &sc, const rtamd_render_params &p,
		const int32_t *glibc_table, double *rays, hipStream_t stream)
{
	TileMap tm = make_tile_map(sc.width, sc.height, p);
	unsigned long long samples = (unsigned long long)tm.owned * (unsigned)(tm.tile_w * tm.tile_h) * (unsigned)sc.spp;
	if (!samples)
		return hipSuccess;
	unsigned long long blocks = (samples + 255) / 256;
	hipLaunchKernelGGL(k_camera_rays, dim3((unsigned)blocks), dim3(256), 0, stream, sc, tm, samples, p.rng_mode,
			p.seed, glibc_table, rays);
	return hipGetLastError();
}

hipError_t rt_launch_tiles_copy(int width, int height, const rtamd_render_params &p, float *frame,
		float *packed, int unpack, hipStream_t stream)
{
	TileMap tm = make_tile_map(width, height, p);
	unsigned long long pixels = (unsigned long long)tm.owned * (unsigned)(tm.tile_w * tm.tile_h);
	if (!pixels)
		return hipSuccess;
	unsigned long long blocks = (pixels + 255) / 256;
	hipLaunchKernelGGL(k_tiles_copy, dim3((unsigned)blocks), dim3(256), 0, stream, width, height, tm, frame,
			packed, unpack);
	return hipGetLastError();
}
