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

// the counter copy this workgroup adds its ray counts to
DEV DevCounters *counters_of_block(DevCounters *counters)
{
	return counters + (blockIdx.x & (RT_COUNTER_COPIES - 1));
}

// entries of the slice that exist
DEV unsigned slice_count(const QueueSlice &q)
{
	unsigned total = *q.count;
	if (total > q.capacity)
		total = q.capacity;
	unsigned n = total > q.first ? total - q.first : 0;
	return n < q.limit ? n : q.limit;
}

// Primary samples: camera ray, nearest hit, direct lighting with shadow rays,
// environment lookup; children go to the bounce queue.  raytracer.c:32-54,78-84.
// Register budget: the primary kernel exists for four waves per SIMD (128 VGPRs, 16
// spilled dwords per lane) and for five (96 VGPRs, 56 spilled dwords).  The kernel is
// bound by the latency of a wave's own instruction stream as much as by the vector
// issue rate: three waves measure 16 % slower than four on the 4K showcase frame, and
// removing 3 % of the vector instructions changes nothing.  Five waves pay where a
// sample's shadow loops are long (the showcase with 17 shadow rays per hit -5 %, 6-soft
// with 48 -5 %, 7-reflections with 50 -6 %, at the price of 150 GB of spill traffic per
// 4K frame) and cost where they are short (3-aa +3 %, the five planes of cornell +6 %),
// six waves are slower everywhere (showcase +9 %); the launcher takes the five-wave form
// from RT_LONG_SHADOW_LOOPS shadow rays per hit in scenes with area lights.  The bounce
// kernels keep per-lane stacks in LDS (32 KB per workgroup) and run at four.
#ifndef RT_MIN_WAVES
#define RT_MIN_WAVES 4
#endif
#ifndef RT_MIN_WAVES_LONG
#define RT_MIN_WAVES_LONG 5
#endif
#ifndef RT_LONG_SHADOW_LOOPS
#define RT_LONG_SHADOW_LOOPS 16
#endif
#ifndef RT_MIN_WAVES_BOUNCE
#define RT_MIN_WAVES_BOUNCE 4
#endif

template <int WAVES>
__global__ void __launch_bounds__(RT_BLOCK, WAVES)
k_primary(DevScene sc, TileMap tm, unsigned long long sample_first, unsigned long long sample_count,
		int rng_mode, uint32_t seed, GlibcTable glibc_table, long long *accum, float *direct_out, float *fsum,
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
	bool active = local < sample_count && sample_to_pixel(tm, sc.sub_count, sample, px, py, k, slot);
	k = sc.sub_first + sc.sub_stride * k;

	uint32_t key = 0;
	Ray ray;
	ray.o = { 0, 0, 0 }; ray.d = { 0, 0, 1 }; ray.tnear = 0; ray.tfar = 0;
	if (active) {
		float offx, offy;
		sample_offsets(sc, rng_mode, glibc_table, seed, px, py, k, key, offx, offy);
		ray = camera_ray(sc, px, py, offx, offy);
	}
	__shared__ ShaftEntry s_shaft[RT_BLOCK / 64][RT_SHAFT_POOL];
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
		sample_to_pixel(tm, sc.sub_count, sample_first + (unsigned long long)blockIdx.x * RT_BLOCK + tid, qx, qy, qk, qslot);
		qk = sc.sub_first + sc.sub_stride * qk;
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
		tactive = l2 < sample_count && sample_to_pixel(tm, sc.sub_count, sample_first + l2, tx, ty, tk, tslot);
	}

	TM_BEGIN(t_tail);
	const unsigned lane = lane_id();
	if (64 % sc.spp == 0 && fsum) {
		// Progressive frame: this call's samples of a pixel (a run of sub_count lanes)
		// continue the pixel's float sum where the pass before left it, in sample order.
		const unsigned run = (unsigned)sc.sub_count;
		const unsigned base = lane - lane % run;
		float4 *cell = reinterpret_cast<float4 *>(fsum) + tslot;
		float ar = 0.0f, ag = 0.0f, ab = 0.0f;
		if (tactive) {
			const float4 before = *cell;      // zero at the start of a frame
			ar = before.x; ag = before.y; ab = before.z;
		}
		for (unsigned j = 0; j < run; j++) {
			ar += __shfl(total.r, (int)(base + j), 64);
			ag += __shfl(total.g, (int)(base + j), 64);
			ab += __shfl(total.b, (int)(base + j), 64);
		}
		if (tactive && lane == base)
			*cell = make_float4(ar, ag, ab, 0.0f);
	} else if (64 % sc.spp == 0) {
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
		// Per-pixel sum in 2^-32 fixed point, every sample converted on its own: integer
		// adds commute, so the result does not depend on the order in which lanes, waves,
		// later bounces or the passes of a progressive frame arrive.
		accumulate_runs(accum, tslot, total, tactive);
	}
	const u64 active_lanes = __builtin_amdgcn_ballot_w64(tactive), env_lanes = __builtin_amdgcn_ballot_w64(n_env != 0);
	DevCounters *mine = counters_of_block(counters);
	if (lane == 0) {
		if (active_lanes) atomicAdd(&mine->rays_primary, (unsigned long long)__builtin_popcountll(active_lanes));
		if (env_lanes) atomicAdd(&mine->env_lookups, (unsigned long long)__builtin_popcountll(env_lanes));
	}
	wave_count_add(&mine->rays_shadow, n_shadow);
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
// PACKET: consecutive queue entries come from the lanes of one spawning wave (a few
// neighbouring pixels' samples).  Mirror reflections of neighbouring samples leave a
// surface from nearly the same point in similar directions: they walk the BVH as a
// packet (4-environmentmap 1024 x 1024 x 64: bounce pass 6.3 -> 4.8 ms).  Glossy
// children are scattered over a cone per sample and walk per lane (packets measured
// three times slower on 8-glossy), with the light loops transposed.
template <bool PACKET>
__global__ void __launch_bounds__(RT_BLOCK, RT_MIN_WAVES_BOUNCE)
k_bounce(DevScene sc, QueueSlice in, unsigned queue_capacity,
		long long *accum, QueueItem *queue_out, unsigned *queue_out_count, DevCounters *counters)
{
	const QueueItem *queue_in = in.items + in.first;
	// packets: a stack and a pool of shaft leaves per wave; per lane: a stack per lane
	__shared__ int s_stack[PACKET ? (RT_BLOCK / 64) * RT_MAX_STACK * 4 : RT_MAX_STACK * RT_BLOCK];
	__shared__ ShaftEntry s_shaft[PACKET ? RT_BLOCK / 64 : 1][PACKET ? RT_SHAFT_POOL : 1];
	int *stack = PACKET ? s_stack + (threadIdx.x >> 6) * RT_MAX_STACK * 4 : s_stack + threadIdx.x;
	ShaftEntry *shaft_lists = PACKET ? s_shaft[threadIdx.x >> 6] : nullptr;
#ifdef RT_TIMING
	if ((threadIdx.x & 63) < TM_COUNT)
		s_timing[threadIdx.x >> 6][threadIdx.x & 63] = 0;
#endif
	TM_BEGIN(t_kernel);
	const unsigned count = slice_count(in);
	if (blockIdx.x == 0 && threadIdx.x == 0)
		atomicMax(&counters->queue_peak, (unsigned long long)*in.count);
	const unsigned stride = gridDim.x * RT_BLOCK;
	// every lane of a wave runs the same number of iterations (the wave-level
	// compaction inside shade_vertex needs all 64 lanes present)
	const unsigned rounds = (count + stride - 1) / stride;
	// the transposed light loops (per-lane form only): a packet stack and the ordered sums per wave
	__shared__ int4 s_wave_stack[PACKET ? 1 : (RT_BLOCK / 64) * RT_MAX_STACK];
	__shared__ float s_wave_sum[PACKET ? 1 : RT_BLOCK / 64][PACKET ? 1 : 384];
	ShadeIO io = { &sc, RTAMD_RNG_KEYED, queue_out, queue_out_count, queue_capacity, counters, shaft_lists,
		PACKET ? nullptr : reinterpret_cast<int *>(s_wave_stack + (threadIdx.x >> 6) * RT_MAX_STACK),
		PACKET ? nullptr : s_wave_sum[threadIdx.x >> 6] };
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
			if (depth == QUEUE_SLOT_UNUSED)
				active = false;
			else
				n_rays++;
		}
		auto ray_again = [&]() {
			unsigned i = idx;
			asm volatile("" : "+v"(i));
			const double *dq = queue_in[i].d;
			return d3{ dq[0], dq[1], dq[2] };
		};
		c3 total = shade_vertex<PACKET, !PACKET>(io, ray, depth, key, weight, slot, active, stack, nullptr, n_shadow, n_env, ray_again);
		// queue order is pixel order: the lanes of a wave mostly share a pixel
		accumulate_runs(accum, slot, cmul(weight, total), active);
	}
	wave_count_add(&counters_of_block(counters)->rays_secondary, n_rays);
	wave_count_add(&counters_of_block(counters)->rays_shadow, n_shadow);
	wave_count_add(&counters_of_block(counters)->env_lookups, n_env);
	TM_END(TM_KERNEL, t_kernel);
#ifdef RT_TIMING
	if ((threadIdx.x & 63) < TM_COUNT)
		atomicAdd(&g_timing_bounce[threadIdx.x & 63], s_timing[threadIdx.x >> 6][threadIdx.x & 63]);
#endif
}

// -----------------------------------------------------------------------------
// Coherent bounce pass.  A bounced hit sends n x n shadow rays to every area
// light, and those are what a reflective frame spends its time on.  The rays
// of one queue wave end all over the scene, so their shadow rays share nothing;
// the shadow rays of hit points that lie NEXT to each other share almost
// everything (the same shaft towards the light, the same few BVH leaves).  So a
// bounce level runs in three steps:
//   k_bounce_trace   one lane per queued ray: nearest hit with a per-lane BVH
//                    walk; misses are shaded here (environment / background);
//                    the bounds of all hit points are collected;
//   k_hit_keys, k_sort_*
//                    the hits, compacted, are radix-sorted by the Morton code of
//                    their position in a grid over those bounds;
//   k_bounce_shade   one lane per hit in sorted order: a wave holds 64
//                    neighbouring hit points whose shadow rays, each walked by
//                    its own lane, visit the same nodes and triangles at the
//                    same time (no divergence to speak of, one cache line per
//                    fetch); children go to the next queue.
// What a lane computes for its hit does not depend on which other hits share
// its wave, and pixel sums are fixed-point integers, so the image is
// bit-identical to the unsorted pass.
// -----------------------------------------------------------------------------

DEV unsigned float_to_ordered(float f)
{
	const unsigned u = (unsigned)as_i(f);
	return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
DEV float ordered_to_float(unsigned e)
{
	return as_f((int)((e & 0x80000000u) ? (e & 0x7fffffffu) : ~e));
}

// bounds[0..2] = min x y z (ordered encoding, atomicMin), bounds[3..5] = ~max
// (atomicMin as well, so that one 0xff fill initialises all six).
__global__ void __launch_bounds__(RT_BLOCK, RT_MIN_WAVES_BOUNCE)
k_bounce_trace(DevScene sc, QueueSlice in, long long *accum, BounceHit *bhit, unsigned *bounds,
		DevCounters *counters)
{
	__shared__ int s_stack[RT_MAX_STACK * RT_BLOCK];
	int *stack = s_stack + threadIdx.x;
	const QueueItem *queue_in = in.items + in.first;
	const unsigned count = slice_count(in);
	if (blockIdx.x == 0 && threadIdx.x == 0)
		atomicMax(&counters->queue_peak, (unsigned long long)*in.count);
	unsigned n_env = 0, n_rays = 0;
	const float inf = __builtin_huge_valf();
	float lo[3] = { inf, inf, inf }, hi[3] = { -inf, -inf, -inf };
	const unsigned stride = gridDim.x * RT_BLOCK;
	const unsigned rounds = (count + stride - 1) / stride;
	for (unsigned round = 0; round < rounds; round++) {
		const unsigned idx = round * stride + blockIdx.x * RT_BLOCK + threadIdx.x;
		const bool present = idx < count;
		QueueItem it;
		it.depth = QUEUE_SLOT_UNUSED;
		it.slot = 0;
		if (present)
			it = queue_in[idx];
		BounceHit h;
		h.surface = -1; h.tri = -1; h.ncode = 0; h.t = 0; h.a = h.b = h.c = 0; h.pad = 0;
		bool missed = false;
		c3 escaped = { 0.0f, 0.0f, 0.0f };
		if (it.depth != QUEUE_SLOT_UNUSED) {
			Ray ray;
			ray.o = { it.o[0], it.o[1], it.o[2] };
			ray.d = { it.d[0], it.d[1], it.d[2] };
			ray.tnear = 0;
			ray.tfar = inf;
			n_rays++;
			Best best;
			if (trace<false, false>(sc, ray, stack, best, shaft_none())) {
				h.surface = best.surface; h.tri = best.tri; h.ncode = best.ncode; h.t = best.t;
				h.a = best.a; h.b = best.b; h.c = best.c;
				const d3 p = vadd(ray.o, vscale((double)best.t, ray.d));
				const float pf[3] = { (float)p.x, (float)p.y, (float)p.z };
				for (int a = 0; a < 3; a++) {
					lo[a] = __builtin_fminf(lo[a], pf[a]);
					hi[a] = __builtin_fmaxf(hi[a], pf[a]);
				}
			} else {
				// shading.c:123-129
				c3 total;
				if (sc.has_cubemap) {
					total = cubemap_colour(sc, ray.d);
					n_env++;
				} else
					total = lc3(sc.background);
				missed = true;
				escaped = cmul({ it.weight[0], it.weight[1], it.weight[2] }, total);
			}
		}
		accumulate_runs(accum, it.slot, escaped, missed);
		if (present)
			bhit[idx] = h;
	}
	for (int a = 0; a < 3; a++) {
		for (int off = 32; off > 0; off >>= 1) {
			lo[a] = __builtin_fminf(lo[a], __shfl_xor(lo[a], off, 64));
			hi[a] = __builtin_fmaxf(hi[a], __shfl_xor(hi[a], off, 64));
		}
		if (lane_id() == 0 && lo[a] <= hi[a]) {
			atomicMin(&bounds[a], float_to_ordered(lo[a]));
			atomicMin(&bounds[3 + a], ~float_to_ordered(hi[a]));
		}
	}
	wave_count_add(&counters_of_block(counters)->rays_secondary, n_rays);
	wave_count_add(&counters_of_block(counters)->env_lookups, n_env);
}

DEV unsigned spread3(unsigned v)
{
	// 10 bits -> every third bit
	v = (v | (v << 16)) & 0x030000ffu;
	v = (v | (v << 8)) & 0x0300f00fu;
	v = (v | (v << 4)) & 0x030c30c3u;
	v = (v | (v << 2)) & 0x09249249u;
	return v;
}

// ---- spatial order of a level's hits -----------------------------------------------
//
// The hits are compacted into (key, queue index) pairs and sorted by key: the Morton
// code of the hit point's cell in a (2^bits)^3 grid over the bounds of this level's hit
// points (up to 10 bits per axis: cells of a hundredth of a unit, the footprint of a
// pixel, so that 64 consecutive hits of a dense level lie as close together as the 64
// samples of a primary pixel).  Least-significant-digit radix sort, RT_SORT_DIGIT bits
// per pass; the number of pairs is only known on the device, so every kernel is
// launched over a fixed grid and derives its share from the counter.  The unit of work
// is a "wave-chunk": a contiguous run of pairs that one wave walks in order, which is
// what makes the scatter stable.
//
// bounds[7] holds the number of hits minus one (all eight words start at 0xffffffff).

DEV unsigned sort_pairs(const unsigned *bounds) { return bounds[7] + 1u; }

// wave-chunks in use and pairs per chunk (a multiple of 64) for n pairs
DEV void sort_chunks(unsigned n, unsigned w_max, unsigned &chunks, unsigned &per)
{
	// at least 1024 pairs to a chunk, at most w_max chunks
	unsigned c = (n + 1023u) / 1024u;
	if (c > w_max) c = w_max;
	if (c < 1u) c = 1u;
	per = (n + c - 1u) / c;
	per = (per + 63u) & ~63u;
	if (per < 64u) per = 64u;
	chunks = (n + per - 1u) / per;
}

__global__ void __launch_bounds__(256)
k_hit_keys(QueueSlice in, const BounceHit *bhit, unsigned *bounds, unsigned *keys, unsigned *index, int bits)
{
	__shared__ unsigned s_wave_hits[4], s_block_base;
	const QueueItem *queue_in = in.items + in.first;
	const unsigned count = slice_count(in);
	const unsigned cells = 1u << bits;
	float lo[3], scale[3];
	for (int a = 0; a < 3; a++) {
		lo[a] = ordered_to_float(bounds[a]);
		const float hi = ordered_to_float(~bounds[3 + a]);
		const float w = hi - lo[a];
		scale[a] = (w > 0.0f && w < 3.0e38f) ? (float)cells / w : 0.0f;
	}
	const unsigned stride = gridDim.x * blockDim.x;
	const unsigned rounds = (count + stride - 1) / stride;
	for (unsigned round = 0; round < rounds; round++) {
		const unsigned idx = round * stride + blockIdx.x * blockDim.x + threadIdx.x;
		bool hit = false;
		unsigned key = 0;
		if (idx < count) {
			const BounceHit h = bhit[idx];
			if (h.surface >= 0) {
				hit = true;
				const QueueItem &it = queue_in[idx];
				unsigned c[3];
				for (int a = 0; a < 3; a++) {
					const float p = (float)(it.o[a] + (double)h.t * it.d[a]);
					const float q = __builtin_fminf(__builtin_fmaxf((p - lo[a]) * scale[a], 0.0f), (float)(cells - 1));
					c[a] = (unsigned)q;
				}
				key = spread3(c[0]) | (spread3(c[1]) << 1) | (spread3(c[2]) << 2);
			}
		}
		// one reservation per workgroup and round (a frame's levels hold tens of millions
		// of hits: one atomic per wave on the one counter would take longer than the rest)
		const u64 hits = __builtin_amdgcn_ballot_w64(hit);
		const unsigned lane = lane_id(), wave = threadIdx.x >> 6;
		if (lane == 0)
			s_wave_hits[wave] = (unsigned)__builtin_popcountll(hits);
		__syncthreads();
		if (threadIdx.x == 0) {
			unsigned total = 0;
			for (unsigned w = 0; w < 4; w++) {
				const unsigned c = s_wave_hits[w];
				s_wave_hits[w] = total;
				total += c;
			}
			s_block_base = total ? atomicAdd(&bounds[7], total) + 1u : 0u;
		}
		__syncthreads();
		if (hit) {
			const unsigned at = s_block_base + s_wave_hits[wave] + (unsigned)__builtin_popcountll(hits & ((1ull << lane) - 1ull));
			keys[at] = key;
			index[at] = idx;
		}
		__syncthreads();
	}
}

DEV unsigned block_exclusive_scan(unsigned v, unsigned *s_wave, unsigned &block_total)
{
	const unsigned lane = lane_id(), wave = threadIdx.x >> 6;
	unsigned incl = v;
	for (int off = 1; off < 64; off <<= 1) {
		const unsigned o = __shfl_up(incl, off, 64);
		if (lane >= (unsigned)off)
			incl += o;
	}
	if (lane == 63)
		s_wave[wave] = incl;
	__syncthreads();
	unsigned before = 0, total = 0;
	for (unsigned w = 0; w < blockDim.x / 64; w++) {
		const unsigned t = s_wave[w];
		if (w < wave)
			before += t;
		total += t;
	}
	__syncthreads();
	block_total = total;
	return before + incl - v;
}

// counts[digit * w_max + chunk] = pairs of the chunk with that digit
__global__ void __launch_bounds__(256)
k_sort_count(const unsigned *keys, const unsigned *bounds, unsigned w_max, int shift, unsigned *counts)
{
	__shared__ unsigned s_cnt[4][RT_SORT_RADIX];
	const unsigned n = sort_pairs(bounds);
	unsigned chunks, per;
	sort_chunks(n, w_max, chunks, per);
	const unsigned wave = threadIdx.x >> 6, lane = lane_id();
	const unsigned chunk = blockIdx.x * 4u + wave;
	if (chunk >= chunks || n == 0)
		return;
	for (unsigned d = lane; d < RT_SORT_RADIX; d += 64)
		s_cnt[wave][d] = 0;
	const unsigned first = chunk * per, last = first + per < n ? first + per : n;
	for (unsigned i = first + lane; i < last; i += 64)
		atomicAdd(&s_cnt[wave][(keys[i] >> shift) & (RT_SORT_RADIX - 1)], 1u);
	for (unsigned d = lane; d < RT_SORT_RADIX; d += 64)
		counts[(size_t)d * w_max + chunk] = s_cnt[wave][d];
}

// One workgroup per digit: exclusive scan of the digit's chunk counts in place, and
// the digit's total.
__global__ void __launch_bounds__(256)
k_sort_scan(unsigned *counts, const unsigned *bounds, unsigned w_max, unsigned *totals)
{
	__shared__ unsigned s_wave[4];
	const unsigned n = sort_pairs(bounds);
	unsigned chunks, per;
	sort_chunks(n, w_max, chunks, per);
	unsigned *row = counts + (size_t)blockIdx.x * w_max;
	const unsigned each = (chunks + 255u) / 256u;
	const unsigned first = threadIdx.x * each;
	unsigned mine = 0;
	for (unsigned k = 0; k < each; k++)
		if (first + k < chunks && n)
			mine += row[first + k];
	unsigned total;
	unsigned run = block_exclusive_scan(mine, s_wave, total);
	for (unsigned k = 0; k < each; k++)
		if (first + k < chunks && n) {
			const unsigned c = row[first + k];
			row[first + k] = run;
			run += c;
		}
	if (threadIdx.x == 0)
		totals[blockIdx.x] = n ? total : 0;
}

// Stable scatter: a wave walks its chunk in order, 64 pairs at a time; a pair's rank
// among the equal digits of its tile is the number of lower lanes with that digit
// (one ballot per digit bit), the digit's running position lives in LDS.
__global__ void __launch_bounds__(256)
k_sort_scatter(const unsigned *keys_in, const unsigned *index_in, const unsigned *bounds, unsigned w_max, int shift,
		const unsigned *counts, const unsigned *totals, unsigned *keys_out, unsigned *index_out)
{
	__shared__ unsigned s_next[4][RT_SORT_RADIX];
	__shared__ unsigned s_base[RT_SORT_RADIX];
	__shared__ unsigned s_wave[4];
	const unsigned n = sort_pairs(bounds);
	unsigned chunks, per;
	sort_chunks(n, w_max, chunks, per);
	if (n == 0 || blockIdx.x * 4u >= chunks)
		return;
	// exclusive scan of the digit totals: where each digit's pairs start
	{
		const unsigned each = RT_SORT_RADIX / 256;
		unsigned v[each], mine = 0;
		for (unsigned k = 0; k < each; k++) {
			v[k] = totals[threadIdx.x * each + k];
			mine += v[k];
		}
		unsigned total;
		unsigned run = block_exclusive_scan(mine, s_wave, total);
		for (unsigned k = 0; k < each; k++) {
			s_base[threadIdx.x * each + k] = run;
			run += v[k];
		}
	}
	__syncthreads();
	const unsigned wave = threadIdx.x >> 6, lane = lane_id();
	const unsigned chunk = blockIdx.x * 4u + wave;
	if (chunk >= chunks)
		return;
	for (unsigned d = lane; d < RT_SORT_RADIX; d += 64)
		s_next[wave][d] = s_base[d] + counts[(size_t)d * w_max + chunk];
	const unsigned first = chunk * per, last = first + per < n ? first + per : n;
	const u64 below = (1ull << lane) - 1ull;
	for (unsigned base = first; base < last; base += 64) {
		const unsigned i = base + lane;
		const bool valid = i < last;
		const unsigned key = valid ? keys_in[i] : 0u, idx = valid ? index_in[i] : 0u;
		const unsigned digit = (key >> shift) & (RT_SORT_RADIX - 1);
		u64 same = __builtin_amdgcn_ballot_w64(valid);
		for (int k = 0; k < RT_SORT_DIGIT; k++) {
			const bool bit = (digit >> k) & 1u;
			const u64 with = __builtin_amdgcn_ballot_w64(bit);
			same &= bit ? with : ~with;
		}
		if (valid) {
			const unsigned start = s_next[wave][digit];
			const unsigned rank = (unsigned)__builtin_popcountll(same & below);
			const unsigned dest = start + rank;
			// the last lane of the group moves the digit on (all lanes of the group have read it:
			// a wave's LDS accesses complete in program order)
			if (rank + 1u == (unsigned)__builtin_popcountll(same))
				s_next[wave][digit] = start + rank + 1u;
			keys_out[dest] = key;
			index_out[dest] = idx;
		}
	}
}

// `n_hits` points at the number of hits minus one (the counter k_hit_keys leaves).
// Shadow rays are walked per lane: measured on 8-glossy (640^2, 64 spp) 384 ms per
// frame against 451 ms with wave-wide shafts and packets (the shaft of 64 distinct
// hit points next to a mesh holds too many leaves, and a packet of their rays is
// the union of 64 pyramids), 451 ms with the transposed loops on sorted hits, and
// 552 ms for the unsorted pass.
// It takes a level only when the level holds at least `min_hits` hits; smaller
// levels go to k_bounce_shade_small.
// the packet form keeps its stacks per wave (14 KB of LDS per workgroup): measured on
// 8-glossy 640 x 640 x 64 with 3 / 4 / 5 / 6 waves per SIMD: 206 / 192 / 186 / 184 ms
#ifndef RT_MIN_WAVES_SHADE_PACKET
#define RT_MIN_WAVES_SHADE_PACKET 5
#endif
template <bool PACKET>
__global__ void __launch_bounds__(RT_BLOCK, PACKET ? RT_MIN_WAVES_SHADE_PACKET : RT_MIN_WAVES_BOUNCE)
k_bounce_shade(DevScene sc, const QueueItem *queue_in, const unsigned *n_hits, unsigned min_hits, unsigned max_hits,
		const BounceHit *bhit, const unsigned *perm, unsigned queue_capacity, long long *accum, QueueItem *queue_out,
		unsigned *queue_out_count, DevCounters *counters, unsigned *cursor)
{
	__shared__ unsigned s_base;
	// per-lane walks: a stack per lane; packets: a stack and a pool of shaft leaves per wave
	__shared__ int s_stack[PACKET ? (RT_BLOCK / 64) * RT_MAX_STACK * 4 : RT_MAX_STACK * RT_BLOCK];
	__shared__ ShaftEntry s_shaft[PACKET ? RT_BLOCK / 64 : 1][PACKET ? RT_SHAFT_POOL : 1];
	int *stack = PACKET ? s_stack + (threadIdx.x >> 6) * RT_MAX_STACK * 4 : s_stack + threadIdx.x;
#ifdef RT_TIMING
	if ((threadIdx.x & 63) < TM_COUNT)
		s_timing[threadIdx.x >> 6][threadIdx.x & 63] = 0;
#endif
	TM_BEGIN(t_kernel);
	unsigned count = *n_hits + 1u;
	if (count > queue_capacity)
		count = queue_capacity;
	if (count < min_hits || count >= max_hits)
		return;
	ShadeIO io = { &sc, RTAMD_RNG_KEYED, queue_out, queue_out_count, queue_capacity, counters,
		PACKET ? s_shaft[threadIdx.x >> 6] : nullptr, nullptr, nullptr };
	unsigned n_shadow = 0;
	// Hit points differ widely in what their shadow rays cost (open floor against the
	// inside of a mesh): workgroups fetch batches of RT_BLOCK hits from a shared cursor
	// instead of owning a fixed share (measured: 310 -> 287 ms on 8-glossy; batches of 64
	// per wave 305 ms, the four waves of a workgroup then work in different places).
	// The cursor starts at 0xffffffff (the fill that initialises the bounds), so the
	// first batch is old + 1 = 0.
	for (;;) {
		__syncthreads();
		if (threadIdx.x == 0)
			s_base = atomicAdd(cursor, (unsigned)RT_BLOCK) + 1u;
		__syncthreads();
		const unsigned base = s_base;
		if (base >= count)
			break;
		const unsigned i = base + threadIdx.x;
		const bool active = i < count;
		unsigned idx = 0;
		d3 ray_d = { 0, 0, 1 };
		c3 weight = { 0, 0, 0 };
		uint32_t slot = 0, key = 0, depth = 0;
		int surface = 0;
		d3 position = { 0, 0, 0 }, normal = { 0, 0, 0 };
		if (active) {
			idx = perm[i];
			const QueueItem it = queue_in[idx];
			const BounceHit h = bhit[idx];
			Ray ray;
			ray.o = { it.o[0], it.o[1], it.o[2] };
			ray.d = { it.d[0], it.d[1], it.d[2] };
			ray.tnear = 0;
			ray.tfar = __builtin_huge_valf();
			ray_d = ray.d;
			weight = { it.weight[0], it.weight[1], it.weight[2] };
			slot = it.slot; key = it.key; depth = it.depth;
			Best best;
			best.surface = h.surface; best.tri = h.tri; best.ncode = h.ncode; best.t = h.t;
			best.a = h.a; best.b = h.b; best.c = h.c;
			surface = h.surface;
			finish_hit(sc, ray, best, position, normal);
		}
		auto ray_again = [&]() {
			unsigned q = idx;
			asm volatile("" : "+v"(q));
			const double *dq = queue_in[q].d;
			return d3{ dq[0], dq[1], dq[2] };
		};
		const c3 total = shade_hit<PACKET, false>(io, active, surface, position, normal, ray_d, depth, key, weight, slot,
				stack, n_shadow, ray_again);
		accumulate_runs(accum, slot, cmul(weight, total), active);
	}
	wave_count_add(&counters_of_block(counters)->rays_shadow, n_shadow);
	TM_END(TM_KERNEL, t_kernel);
#ifdef RT_TIMING
	if ((threadIdx.x & 63) < TM_COUNT)
		atomicAdd(&g_timing_bounce[threadIdx.x & 63], s_timing[threadIdx.x >> 6][threadIdx.x & 63]);
#endif
}

// The same step for a level of fewer than `max_hits` hits.  One lane per hit would
// leave most of the GPU idle while a few waves walk 2 n^2 shadow rays per lane one
// after the other (a level of any size then takes as long as a lane takes).  Here
// a wave takes only as many hits as have their light samples side by side in its
// 64 lanes (lights_transposed: lane = (hit, sample), one packet walk per light),
// so a small level spreads over many waves and finishes in the time of a few
// packet walks.  Same sums per hit, in the same order.
__global__ void __launch_bounds__(RT_BLOCK, RT_MIN_WAVES_BOUNCE)
k_bounce_shade_small(DevScene sc, const QueueItem *queue_in, const unsigned *n_hits, unsigned max_hits,
		unsigned hits_per_wave, const BounceHit *bhit, const unsigned *perm, unsigned queue_capacity, long long *accum,
		QueueItem *queue_out, unsigned *queue_out_count, DevCounters *counters)
{
	__shared__ int s_stack[RT_MAX_STACK * RT_BLOCK];
	__shared__ int4 s_wave_stack[(RT_BLOCK / 64) * RT_MAX_STACK];
	__shared__ float s_wave_sum[RT_BLOCK / 64][384];
	int *stack = s_stack + threadIdx.x;
	unsigned count = *n_hits + 1u;
	if (count > queue_capacity)
		count = queue_capacity;
	if (count >= max_hits)
		return;
	const unsigned wave_in_block = threadIdx.x >> 6, lane = lane_id();
	const unsigned waves = gridDim.x * (RT_BLOCK / 64);
	const unsigned groups = (count + hits_per_wave - 1) / hits_per_wave;
	const unsigned rounds = (groups + waves - 1) / waves;
	ShadeIO io = { &sc, RTAMD_RNG_KEYED, queue_out, queue_out_count, queue_capacity, counters, nullptr,
		reinterpret_cast<int *>(s_wave_stack + wave_in_block * RT_MAX_STACK), s_wave_sum[wave_in_block] };
	unsigned n_shadow = 0;
	for (unsigned round = 0; round < rounds; round++) {
		const unsigned group = round * waves + blockIdx.x * (RT_BLOCK / 64) + wave_in_block;
		const unsigned i = group * hits_per_wave + lane;
		const bool active = lane < hits_per_wave && group < groups && i < count;
		unsigned idx = 0;
		d3 ray_d = { 0, 0, 1 };
		c3 weight = { 0, 0, 0 };
		uint32_t slot = 0, key = 0, depth = 0;
		int surface = 0;
		d3 position = { 0, 0, 0 }, normal = { 0, 0, 0 };
		if (active) {
			idx = perm[i];
			const QueueItem it = queue_in[idx];
			const BounceHit h = bhit[idx];
			Ray ray;
			ray.o = { it.o[0], it.o[1], it.o[2] };
			ray.d = { it.d[0], it.d[1], it.d[2] };
			ray.tnear = 0;
			ray.tfar = __builtin_huge_valf();
			ray_d = ray.d;
			weight = { it.weight[0], it.weight[1], it.weight[2] };
			slot = it.slot; key = it.key; depth = it.depth;
			Best best;
			best.surface = h.surface; best.tri = h.tri; best.ncode = h.ncode; best.t = h.t;
			best.a = h.a; best.b = h.b; best.c = h.c;
			surface = h.surface;
			finish_hit(sc, ray, best, position, normal);
		}
		auto ray_again = [&]() {
			unsigned q = idx;
			asm volatile("" : "+v"(q));
			const double *dq = queue_in[q].d;
			return d3{ dq[0], dq[1], dq[2] };
		};
		const c3 total = shade_hit<false, true>(io, active, surface, position, normal, ray_d, depth, key, weight, slot,
				stack, n_shadow, ray_again);
		accumulate_runs(accum, slot, cmul(weight, total), active);
	}
	wave_count_add(&counters_of_block(counters)->rays_shadow, n_shadow);
}

// Folds the counter copies into copy 0 (one thread per copy) and clears them.
__global__ void __launch_bounds__(RT_COUNTER_COPIES)
k_collect_stats(DevCounters *counters)
{
	if (threadIdx.x > 0) {
		DevCounters *c = counters + threadIdx.x;
		if (c->rays_primary) atomicAdd(&counters->rays_primary, c->rays_primary);
		if (c->rays_shadow) atomicAdd(&counters->rays_shadow, c->rays_shadow);
		if (c->rays_secondary) atomicAdd(&counters->rays_secondary, c->rays_secondary);
		if (c->env_lookups) atomicAdd(&counters->env_lookups, c->env_lookups);
		c->rays_primary = c->rays_shadow = c->rays_secondary = c->env_lookups = 0;
		return;
	}
#ifdef RT_TRAVERSAL_STATS
	counters->bvh_nodes = g_stat_nodes;
	counters->tri_tests = g_stat_tris;
	g_stat_nodes = 0;
	g_stat_tris = 0;
	for (int i = 0; i < RT_SHAFT_CAP + 16; i++) {
		if (g_stat_shaft[i])
			printf("shaft %d %llu\n", i, g_stat_shaft[i]);
		g_stat_shaft[i] = 0;
	}
	for (int i = 32; i < 40; i++) {
		if (g_stat_ev[i])
			printf("event %d %llu\n", i, g_stat_ev[i]);
		g_stat_ev[i] = 0;
	}
	for (int i = 0; i < 32; i++) {
		if (g_stat_ev[i])
			printf("event %d %llu\n", i, g_stat_ev[i]);
		g_stat_ev[i] = 0;
		if (g_stat_walk[i])
			printf("walk %s steps<2^%d %llu\n", i & 1 ? "occluded" : "free", i / 2, g_stat_walk[i]);
		g_stat_walk[i] = 0;
		if (i < 8 && g_stat_list[i])
			printf("list %d %llu\n", i, g_stat_list[i]);
		if (i < 8)
			g_stat_list[i] = 0;
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

// The reference's rand() stream (ray.c:14-17 after srand, raytracer.c:74) for a run
// of draws, generated where it is used.  The generator is linear: its raw values
// obey w[i+31] = w[i+28] + w[i] (mod 2^32), so with c(x) = x^n mod (x^31 - x^28 - 1)
// the value at n + k is sum_j c[j] w[j + k] over the first values of the sequence.
// A workgroup produces RT_GLIBC_BLOCK consecutive draws: it multiplies the launch's
// starting polynomial by the tabulated powers x^(RT_GLIBC_BLOCK * b) for its own
// block number b (two polynomial products), then every thread combines the window
// w[0 .. RT_GLIBC_BLOCK + 30] with the 31 coefficients.  draw = value >> 1.
DEV void glibc_poly_mul_block(uint32_t *a, const uint32_t *b, uint32_t *e)
{
	// a = a * b mod p; a in LDS, b anywhere, e = 61 words of LDS
	const int k = threadIdx.x;
	if (k < 2 * RT_GLIBC_POLY - 1) {
		uint32_t v = 0;
		const int i0 = k > RT_GLIBC_POLY - 1 ? k - (RT_GLIBC_POLY - 1) : 0;
		const int i1 = k < RT_GLIBC_POLY - 1 ? k : RT_GLIBC_POLY - 1;
		for (int i = i0; i <= i1; i++)
			v += a[i] * b[k - i];
		e[k] = v;
	}
	__syncthreads();
	if (k == 0)
		for (int m = 2 * RT_GLIBC_POLY - 2; m >= RT_GLIBC_POLY; m--) {
			e[m - 3] += e[m];
			e[m - RT_GLIBC_POLY] += e[m];
		}
	__syncthreads();
	if (k < RT_GLIBC_POLY)
		a[k] = e[k];
	__syncthreads();
}

__global__ void __launch_bounds__(256)
k_glibc_fill(GlibcJump start, const uint32_t *powers, const uint32_t *window, int32_t *out, unsigned long long count)
{
	__shared__ uint32_t s_c[RT_GLIBC_POLY], s_e[2 * RT_GLIBC_POLY - 1];
	__shared__ uint32_t s_w[RT_GLIBC_BLOCK + RT_GLIBC_POLY];
	for (int i = threadIdx.x; i < RT_GLIBC_BLOCK + RT_GLIBC_POLY; i += 256)
		s_w[i] = window[i];
	if (threadIdx.x < RT_GLIBC_POLY)
		s_c[threadIdx.x] = start.c[threadIdx.x];
	__syncthreads();
	const unsigned b = blockIdx.x;
	// powers[0 .. 1024) = x^(BLOCK * i), powers[1024 .. 2048) = x^(BLOCK * 1024 * i)
	glibc_poly_mul_block(s_c, powers + (size_t)(1024 + (b >> 10)) * RT_GLIBC_POLY, s_e);
	glibc_poly_mul_block(s_c, powers + (size_t)(b & 1023) * RT_GLIBC_POLY, s_e);
	uint32_t c[RT_GLIBC_POLY];
	for (int j = 0; j < RT_GLIBC_POLY; j++)
		c[j] = s_c[j];
	const unsigned long long base = (unsigned long long)b * RT_GLIBC_BLOCK;
	for (int k = threadIdx.x; k < RT_GLIBC_BLOCK; k += 256) {
		if (base + (unsigned)k >= count)
			break;
		uint32_t v = 0;
		for (int j = 0; j < RT_GLIBC_POLY; j++)
			v += c[j] * s_w[j + k];
		out[base + (unsigned)k] = (int32_t)(v >> 1);
	}
}

hipError_t rt_launch_glibc_fill(const GlibcJump &start, const uint32_t *powers, const uint32_t *window,
		int32_t *out, unsigned long long count, hipStream_t stream)
{
	if (!count)
		return hipSuccess;
	const unsigned long long blocks = (count + RT_GLIBC_BLOCK - 1) / RT_GLIBC_BLOCK;
	hipLaunchKernelGGL(k_glibc_fill, dim3((unsigned)blocks), dim3(256), 0, stream, start, powers, window, out, count);
	return hipGetLastError();
}

// Accumulators -> the reference's float RGBA frame (raytracer.c:46,81): the
// per-pixel sum is scaled by the float 1.0/aa^2 when antialiasing is on.
__global__ void k_resolve(DevScene sc, TileMap tm, unsigned long long pixel_first, unsigned long long pixel_count,
		const long long *accum, const float *fsum, float *colour, const DevCounters *counters,
		unsigned long long *overflow_out)
{
	unsigned long long local = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
	// the last kernel of a reflective frame leaves a queue overflow where the host
	// finds it without waiting for the stream (host-visible memory)
	if (local == 0 && overflow_out && counters->queue_overflow)
		*overflow_out = counters->queue_overflow;
	if (local >= pixel_count)
		return;
	unsigned long long pix = pixel_first + local;
	int px, py, k;
	uint32_t slot;
	if (!sample_to_pixel(tm, 1, pix, px, py, k, slot))
		return;
	// a progressive frame keeps the ordered float sum of the directly lit part beside
	// the fixed-point sums of what the bounces add (the single call adds the former to
	// the latter as one fixed-point value: the same total)
	float r = 0.0f, g = 0.0f, b = 0.0f;
	long long fix[3] = { 0, 0, 0 };
	if (fsum) {
		const float4 f = reinterpret_cast<const float4 *>(fsum)[slot];
		r = f.x; g = f.y; b = f.z;
		fix[0] = to_fixed(f.x); fix[1] = to_fixed(f.y); fix[2] = to_fixed(f.z);
	}
	if (accum) {
		const long long *a = accum + 3ull * slot;
		r = (float)((double)(a[0] + fix[0]) / RT_FIX_SCALE);
		g = (float)((double)(a[1] + fix[1]) / RT_FIX_SCALE);
		b = (float)((double)(a[2] + fix[2]) / RT_FIX_SCALE);
	}
	if (sc.antialiasing) {
		float s = (float)(1.0 / (double)sc.resolve_samples);
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
		uint32_t seed, GlibcTable glibc_table, double *rays)
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

// The tiles of several owners in one launch (rank 0 after the framebuffer gather):
// blockIdx.y selects the owner, whose dense buffer starts owner_stride float4 after
// the previous owner's.
__global__ void k_tiles_unpack_all(int width, int height, TileMap tm, int first_owner, float *frame,
		const float *packed_all, unsigned long long owner_stride)
{
	const int owner = first_owner + (int)blockIdx.y;
	const int total = tm.tiles_x * tm.tiles_y;
	tm.tile_first = owner;
	tm.owned = owner < total ? (total - owner + tm.tile_stride - 1) / tm.tile_stride : 0;
	tm.x0 = 0; tm.y0 = 0; tm.x1 = width; tm.y1 = height;
	unsigned long long pix = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (pix >= (unsigned long long)tm.owned * (unsigned)(tm.tile_w * tm.tile_h))
		return;
	int px, py, k;
	uint32_t slot;
	if (!sample_to_pixel(tm, 1, pix, px, py, k, slot))
		return;
	const float4 *p = reinterpret_cast<const float4 *>(packed_all) + owner_stride * (unsigned long long)owner + pix;
	reinterpret_cast<float4 *>(frame)[(unsigned long long)width * py + px] = *p;
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

// Positions [lo, hi) in the owner's tile sequence of the owned tiles that lie in the
// tile rows [row0, row1) (tiles are numbered row by row, so a run of tile rows is one
// contiguous run of tile numbers).
void rt_tile_rows_range(int width, int height, const rtamd_render_params &p, int row0, int row1,
		unsigned long long *lo, unsigned long long *hi)
{
	const TileMap tm = make_tile_map(width, height, p);
	const long long t0 = (long long)row0 * tm.tiles_x, t1 = (long long)row1 * tm.tiles_x;
	auto position = [&](long long tile) {
		// number of owned tiles below `tile`: ceil((tile - first) / stride), within [0, owned]
		long long n = tile <= tm.tile_first ? 0 : (tile - tm.tile_first + tm.tile_stride - 1) / tm.tile_stride;
		return (unsigned long long)(n > tm.owned ? tm.owned : n);
	};
	*lo = position(t0);
	*hi = position(t1);
}

// The tile rows [row0, row1) that a call's pixel window touches (the whole frame: all rows).
void rt_window_tile_rows(int width, int height, const rtamd_render_params &p, int *row0, int *row1)
{
	const TileMap tm = make_tile_map(width, height, p);
	*row0 = tm.y0 / tm.tile_h;
	*row1 = (tm.y1 - 1) / tm.tile_h + 1;
}

hipError_t rt_launch_primary(const DevScene &sc, const rtamd_render_params &p,
		unsigned long long sample_first, unsigned long long sample_count, GlibcTable glibc_table,
		long long *accum, float *direct_out, float *fsum, rtamd_hit_record *hits, QueueItem *queue_out,
		unsigned *queue_count, unsigned queue_capacity, DevCounters *counters, hipStream_t stream)
{
	if (!sample_count)
		return hipSuccess;
	TileMap tm = make_tile_map(sc.width, sc.height, p);
	unsigned long long blocks = (sample_count + RT_BLOCK - 1) / RT_BLOCK;
	if (sc.any_area_light && sc.shadow_rays_per_hit >= RT_LONG_SHADOW_LOOPS)
		hipLaunchKernelGGL(k_primary<RT_MIN_WAVES_LONG>, dim3((unsigned)blocks), dim3(RT_BLOCK), 0, stream, sc, tm,
				sample_first, sample_count, p.rng_mode, p.seed, glibc_table, accum, direct_out, fsum, hits, queue_out,
				queue_count, queue_capacity, counters);
	else
		hipLaunchKernelGGL(k_primary<RT_MIN_WAVES>, dim3((unsigned)blocks), dim3(RT_BLOCK), 0, stream, sc, tm,
				sample_first, sample_count, p.rng_mode, p.seed, glibc_table, accum, direct_out, fsum, hits, queue_out,
				queue_count, queue_capacity, counters);
	return hipGetLastError();
}

hipError_t rt_launch_bounce(const DevScene &sc, const QueueSlice &in, long long *accum, QueueItem *queue_out,
		unsigned *queue_out_count, DevCounters *counters, unsigned grid_blocks, bool packets, hipStream_t stream)
{
	if (packets)
		hipLaunchKernelGGL(k_bounce<true>, dim3(grid_blocks), dim3(RT_BLOCK), 0, stream, sc, in, in.capacity, accum,
				queue_out, queue_out_count, counters);
	else
		hipLaunchKernelGGL(k_bounce<false>, dim3(grid_blocks), dim3(RT_BLOCK), 0, stream, sc, in, in.capacity, accum,
				queue_out, queue_out_count, counters);
	return hipGetLastError();
}

// One bounce level of the coherent pass.  `work` holds the level's scratch buffers.
hipError_t rt_launch_bounce_sorted(const DevScene &sc, const QueueSlice &in, long long *accum, QueueItem *queue_out,
		unsigned *queue_out_count, DevCounters *counters, const BounceWork &work, unsigned grid_blocks,
		hipStream_t stream)
{
	const int bits = work.bits;
	hipError_t e;
	// hit-point bounds, batch cursor, hit counter: all start at 0xffffffff
	if ((e = hipMemsetAsync(work.bounds, 0xff, 8 * sizeof(unsigned), stream)) != hipSuccess) return e;
	hipLaunchKernelGGL(k_bounce_trace, dim3(grid_blocks), dim3(RT_BLOCK), 0, stream, sc, in, accum, work.hits,
			work.bounds, counters);
	hipLaunchKernelGGL(k_hit_keys, dim3(grid_blocks), dim3(256), 0, stream, in, work.hits, work.bounds, work.keys[0],
			work.index[0], bits);
	const int passes = (3 * bits + RT_SORT_DIGIT - 1) / RT_SORT_DIGIT;
	const unsigned sort_blocks = (work.sort_chunks_max + 3u) / 4u;
	int from = 0;
	for (int pass = 0; pass < passes; pass++) {
		const int shift = pass * RT_SORT_DIGIT;
		hipLaunchKernelGGL(k_sort_count, dim3(sort_blocks), dim3(256), 0, stream, work.keys[from], work.bounds,
				work.sort_chunks_max, shift, work.counts);
		hipLaunchKernelGGL(k_sort_scan, dim3(RT_SORT_RADIX), dim3(256), 0, stream, work.counts, work.bounds,
				work.sort_chunks_max, work.totals);
		hipLaunchKernelGGL(k_sort_scatter, dim3(sort_blocks), dim3(256), 0, stream, work.keys[from], work.index[from],
				work.bounds, work.sort_chunks_max, shift, work.counts, work.totals, work.keys[1 - from],
				work.index[1 - from]);
		from = 1 - from;
	}
	const unsigned *perm = work.index[from];
	// Three kernels share a level by its number of hits: the thinnest levels spread each
	// hit's light samples over a wave, the middle ones walk per lane, dense ones - where 64
	// neighbouring hits lie within a pixel's footprint of each other - in packets.
	const unsigned small_below = work.small_below;
	const unsigned packets_from = work.packets_from > small_below ? work.packets_from : small_below;
	const unsigned *n_hits = work.bounds + 7;
	const QueueItem *items = in.items + in.first;
	if (packets_from > small_below)
		hipLaunchKernelGGL(k_bounce_shade<false>, dim3(grid_blocks), dim3(RT_BLOCK), 0, stream, sc, items, n_hits,
				small_below, packets_from, work.hits, perm, in.capacity, accum, queue_out, queue_out_count, counters,
				work.bounds + 6);
	if (packets_from != 0xffffffffu)
		hipLaunchKernelGGL(k_bounce_shade<true>, dim3(grid_blocks), dim3(RT_BLOCK), 0, stream, sc, items, n_hits,
				packets_from, 0xffffffffu, work.hits, perm, in.capacity, accum, queue_out, queue_out_count, counters,
				work.bounds + 6);
	if (small_below) {
		const unsigned per_wave = work.hits_per_wave;
		unsigned blocks = (small_below / per_wave + RT_BLOCK / 64 - 1) / (RT_BLOCK / 64);
		if (blocks < 1) blocks = 1;
		if (blocks > 65536) blocks = 65536;
		hipLaunchKernelGGL(k_bounce_shade_small, dim3(blocks), dim3(RT_BLOCK), 0, stream, sc, items, n_hits,
				small_below, per_wave, work.hits, perm, in.capacity, accum, queue_out, queue_out_count, counters);
	}
	return hipGetLastError();
}

hipError_t rt_launch_resolve(const DevScene &sc, const rtamd_render_params &p, unsigned long long pixel_first,
		unsigned long long pixel_count, const long long *accum, const float *fsum, float *colour,
		DevCounters *counters, unsigned long long *overflow_out, hipStream_t stream)
{
	if (!pixel_count)
		return hipSuccess;
	TileMap tm = make_tile_map(sc.width, sc.height, p);
	unsigned long long blocks = (pixel_count + 255) / 256;
	hipLaunchKernelGGL(k_resolve, dim3((unsigned)blocks), dim3(256), 0, stream, sc, tm, pixel_first, pixel_count,
			accum, fsum, colour, counters, overflow_out);
	return hipGetLastError();
}

// Folds the counter copies into the first one (and, in statistics builds, moves the
// traversal statistics there).
hipError_t rt_launch_collect_stats(DevCounters *counters, hipStream_t stream)
{
	hipLaunchKernelGGL(k_collect_stats, dim3(1), dim3(RT_COUNTER_COPIES), 0, stream, counters);
	return hipGetLastError();
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
		GlibcTable glibc_table, double *rays, hipStream_t stream)
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

hipError_t rt_launch_tiles_unpack_all(int width, int height, const rtamd_render_params &p, int first_owner,
		float *frame, const float *packed_all, unsigned long long owner_stride_float4, hipStream_t stream)
{
	rtamd_render_params q = p;
	q.tile_first = 0;
	TileMap tm = make_tile_map(width, height, q);
	// owner 0 holds the most tiles
	unsigned long long pixels = (unsigned long long)tm.owned * (unsigned)(tm.tile_w * tm.tile_h);
	const int owners = p.tile_stride - first_owner;
	if (!pixels || owners <= 0)
		return hipSuccess;
	unsigned long long blocks = (pixels + 255) / 256;
	hipLaunchKernelGGL(k_tiles_unpack_all, dim3((unsigned)blocks, (unsigned)owners), dim3(256), 0, stream, width, height,
			tm, first_owner, frame, packed_all, owner_stride_float4);
	return hipGetLastError();
}
