/* Host-callable launchers of the HIP kernels (kernels.hip). Internal. */
#ifndef RTAMD_KERNELS_H
#define RTAMD_KERNELS_H

#include <hip/hip_runtime.h>
#include "scene_dev.h"

int rt_tiles_owned(int width, int height, const rtamd_render_params &p);
unsigned long long rt_owned_pixels(int width, int height, const rtamd_render_params &p);
void rt_tile_rows_range(int width, int height, const rtamd_render_params &p, int row0, int row1,
		unsigned long long *lo, unsigned long long *hi);
void rt_window_tile_rows(int width, int height, const rtamd_render_params &p, int *row0, int *row1);
hipError_t rt_launch_glibc_fill(const GlibcJump &start, const uint32_t *powers, const uint32_t *window,
		int32_t *out, unsigned long long count, hipStream_t stream);

hipError_t rt_launch_primary(const DevScene &sc, const rtamd_render_params &p,
		unsigned long long sample_first, unsigned long long sample_count, GlibcTable glibc_table,
		long long *accum, float *direct_out, float *fsum, rtamd_hit_record *hits, QueueItem *queue_out,
		unsigned *queue_count, unsigned queue_capacity, DevCounters *counters, hipStream_t stream);
/* Scratch of one bounce level of the coherent pass: per queue entry the nearest
 * hit; two buffers of (key, queue index) pairs for the radix sort of the hits; the
 * sort's digit counts per wave-chunk and digit totals; eight words for the hit-point
 * bounds, the batch cursor and the hit counter. */
struct BounceWork {
	BounceHit *hits;
	unsigned *keys[2];
	unsigned *index[2];
	unsigned *counts;        /* RT_SORT_RADIX x sort_chunks_max */
	unsigned *totals;        /* RT_SORT_RADIX */
	unsigned *bounds;
	int bits;                /* grid bits per axis of the sort key */
	unsigned sort_chunks_max;/* wave-chunks the sort may use */
	unsigned small_below;    /* levels of fewer hits are shaded with their samples across lanes; 0 = never */
	unsigned hits_per_wave;  /* ... this many hits to a wave */
	unsigned packets_from;   /* levels of at least this many hits are shaded with wave-wide shafts and packets */
};
hipError_t rt_launch_bounce(const DevScene &sc, const QueueSlice &in, long long *accum, QueueItem *queue_out,
		unsigned *queue_out_count, DevCounters *counters, unsigned grid_blocks, bool packets, hipStream_t stream);
hipError_t rt_launch_bounce_sorted(const DevScene &sc, const QueueSlice &in, long long *accum, QueueItem *queue_out,
		unsigned *queue_out_count, DevCounters *counters, const BounceWork &work, unsigned grid_blocks,
		hipStream_t stream);
hipError_t rt_launch_resolve(const DevScene &sc, const rtamd_render_params &p, unsigned long long pixel_first,
		unsigned long long pixel_count, const long long *accum, const float *fsum, float *colour,
		DevCounters *counters, unsigned long long *overflow_out, hipStream_t stream);
hipError_t rt_launch_collect_stats(DevCounters *counters, hipStream_t stream);
hipError_t rt_launch_trace(const DevScene &sc, const double *rays, unsigned long long n,
		rtamd_hit_record *hits, hipStream_t stream);
hipError_t rt_launch_camera_rays(const DevScene &sc, const rtamd_render_params &p,
		GlibcTable glibc_table, double *rays, hipStream_t stream);
hipError_t rt_launch_tiles_copy(int width, int height, const rtamd_render_params &p, float *frame,
		float *packed, int unpack, hipStream_t stream);
hipError_t rt_launch_tiles_unpack_all(int width, int height, const rtamd_render_params &p, int first_owner,
		float *frame, const float *packed_all, unsigned long long owner_stride_float4, hipStream_t stream);

#endif
