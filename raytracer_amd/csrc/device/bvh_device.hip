// BVH construction on the GPU for one mesh: a linear BVH (Morton order of the
// triangle centroids, radix tree over the sorted codes, boxes fitted bottom-up),
// written in the layout the traversal kernels read: 64-byte two-child nodes,
// leaf-ordered triangles / normals / edge records.
//
// This structure takes the place of the reference's kd-tree build
// (mesh.c:350-451, triggered at scene.c:659-668).  Hit records do not depend on
// the tree's shape (nearest t, ties to the highest triangle index), so the
// device-built tree and the host's SAH tree give identical frames; the SAH tree
// is the better one to traverse, the device build is for meshes whose host build
// would take longer than their first frames.
//
// Steps, all on the caller's stream:
//   k_lbvh_bounds    bounds of the triangle centroids (ordered-float atomics)
//   k_lbvh_codes     30-bit Morton code of every centroid
//   4 x (k_radix_count, k_radix_scan, k_radix_scatter)
//                    stable LSD radix sort of (code, triangle) by 8-bit digits
//   k_lbvh_tree      the radix tree: every inner node finds its range of sorted
//                    triangles and its split from the codes' common prefixes
//                    (equal codes are told apart by their position)
//   k_lbvh_fit       boxes from the leaves upwards; the second child to arrive
//                    at a node computes its box
//   k_lbvh_emit      nodes in the traversal layout; subtrees of at most
//                    RT_MAX_LEAF triangles become leaves
//   k_lbvh_gather    triangles, normals and edge records in sorted order
//   k_lbvh_depth     depth of the tree (the traversal stack bounds it)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bvh_device.h"

#ifndef RT_MAX_LEAF
#define RT_MAX_LEAF 2
#endif

namespace {

#define DEV __device__ __forceinline__

const int RADIX_BITS = 8;
const int RADIX = 1 << RADIX_BITS;
const int SORT_BLOCKS_MAX = 1024;      // blocks of the sort; one wave each

DEV unsigned lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

DEV unsigned float_to_ordered(float f)
{
	const unsigned u = __builtin_bit_cast(unsigned, f);
	return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
DEV float ordered_to_float(unsigned e)
{
	return __builtin_bit_cast(float, (e & 0x80000000u) ? (e & 0x7fffffffu) : ~e);
}

DEV void triangle_centroid(const float *pos, int i, float c[3])
{
	const float *p = pos + 9 * (size_t)i;
	for (int a = 0; a < 3; a++)
		c[a] = (p[a] + p[3 + a] + p[6 + a]) * (1.0f / 3.0f);
}

// bounds[0..2] = min, bounds[3..5] = ~max, both kept with atomicMin (filled with 0xff)
__global__ void k_lbvh_bounds(const float *pos, int n, unsigned *bounds)
{
	const float inf = __builtin_huge_valf();
	float lo[3] = { inf, inf, inf }, hi[3] = { -inf, -inf, -inf };
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		float c[3];
		triangle_centroid(pos, i, c);
		for (int a = 0; a < 3; a++) {
			lo[a] = __builtin_fminf(lo[a], c[a]);
			hi[a] = __builtin_fmaxf(hi[a], c[a]);
		}
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
}

DEV unsigned spread10(unsigned v)
{
	v = (v | (v << 16)) & 0x030000ffu;
	v = (v | (v << 8)) & 0x0300f00fu;
	v = (v | (v << 4)) & 0x030c30c3u;
	v = (v | (v << 2)) & 0x09249249u;
	return v;
}

__global__ void k_lbvh_codes(const float *pos, int n, const unsigned *bounds, unsigned *keys, unsigned *index)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
		return;
	float c[3];
	triangle_centroid(pos, i, c);
	unsigned q[3];
	for (int a = 0; a < 3; a++) {
		const float lo = ordered_to_float(bounds[a]), hi = ordered_to_float(~bounds[3 + a]);
		const float w = hi - lo;
		const float s = (w > 0.0f && w < 3.0e38f) ? 1024.0f / w : 0.0f;
		const float v = __builtin_fminf(__builtin_fmaxf((c[a] - lo) * s, 0.0f), 1023.0f);
		q[a] = (unsigned)v;
	}
	keys[i] = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
	index[i] = (unsigned)i;
}

// One wave per block; block b owns the elements [b * chunk, (b + 1) * chunk).
// counts[digit * blocks + b] = how many of the block's keys have that digit.
__global__ void __launch_bounds__(64)
k_radix_count(const unsigned *keys, int n, int chunk, int shift, unsigned *counts)
{
	__shared__ unsigned s_count[RADIX];
	for (int d = threadIdx.x; d < RADIX; d += 64)
		s_count[d] = 0;
	__syncthreads();
	const int first = blockIdx.x * chunk, last = min(n, first + chunk);
	for (int i = first + (int)threadIdx.x; i < last; i += 64)
		atomicAdd(&s_count[(keys[i] >> shift) & (RADIX - 1)], 1u);
	__syncthreads();
	for (int d = threadIdx.x; d < RADIX; d += 64)
		counts[d * gridDim.x + blockIdx.x] = s_count[d];
}

// Exclusive scan of counts[RADIX * blocks] (digit-major) by one workgroup.
__global__ void __launch_bounds__(1024)
k_radix_scan(unsigned *counts, int total)
{
	__shared__ unsigned s_part[1024];
	const int per = (total + 1023) / 1024;
	const int first = threadIdx.x * per;
	unsigned sum = 0;
	for (int k = 0; k < per; k++)
		if (first + k < total)
			sum += counts[first + k];
	s_part[threadIdx.x] = sum;
	__syncthreads();
	// Hillis-Steele over the 1024 partial sums
	for (int off = 1; off < 1024; off <<= 1) {
		unsigned v = threadIdx.x >= (unsigned)off ? s_part[threadIdx.x - off] : 0;
		__syncthreads();
		s_part[threadIdx.x] += v;
		__syncthreads();
	}
	unsigned run = s_part[threadIdx.x] - sum;
	for (int k = 0; k < per; k++)
		if (first + k < total) {
			const unsigned c = counts[first + k];
			counts[first + k] = run;
			run += c;
		}
}

// Stable scatter: the block walks its chunk in order, 64 elements at a time; inside
// a tile an element's rank among equal digits is the number of lower lanes holding
// the same digit.
__global__ void __launch_bounds__(64)
k_radix_scatter(const unsigned *keys_in, const unsigned *index_in, int n, int chunk, int shift,
		const unsigned *offsets, unsigned *keys_out, unsigned *index_out)
{
	__shared__ unsigned s_next[RADIX];
	for (int d = threadIdx.x; d < RADIX; d += 64)
		s_next[d] = offsets[d * gridDim.x + blockIdx.x];
	__syncthreads();
	const int first = blockIdx.x * chunk, last = min(n, first + chunk);
	const unsigned lane = threadIdx.x;
	for (int base = first; base < last; base += 64) {
		const int i = base + (int)lane;
		const bool valid = i < last;
		const unsigned key = valid ? keys_in[i] : 0, idx = valid ? index_in[i] : 0;
		const unsigned digit = (key >> shift) & (RADIX - 1);
		unsigned dest = 0;
		unsigned long long todo = __builtin_amdgcn_ballot_w64(valid);
		while (todo) {
			const int leader = __builtin_ctzll(todo);
			const unsigned d = (unsigned)__builtin_amdgcn_readlane((int)digit, leader);
			const unsigned long long same = __builtin_amdgcn_ballot_w64(valid && digit == d) & todo;
			const unsigned start = s_next[d];
			if (valid && digit == d)
				dest = start + (unsigned)__builtin_popcountll(same & ((1ull << lane) - 1ull));
			__syncthreads();
			if ((int)lane == leader)
				s_next[d] = start + (unsigned)__builtin_popcountll(same);
			__syncthreads();
			todo &= ~same;
		}
		if (valid) {
			keys_out[dest] = key;
			index_out[dest] = idx;
		}
	}
}

// Length of the common prefix of the sorted keys at positions i and j; equal codes
// are extended by the positions themselves, so all keys are distinct.  -1 outside.
DEV int prefix_len(const unsigned *keys, int n, int i, int j)
{
	if (j < 0 || j >= n)
		return -1;
	const unsigned a = keys[i], b = keys[j];
	if (a != b)
		return __builtin_clz(a ^ b);
	return 32 + __builtin_clz((unsigned)i ^ (unsigned)j);
}

// Inner node i of the radix tree over n sorted keys (n - 1 inner nodes, node 0 is
// the root).  Children are inner nodes (index >= 0) or single sorted triangles
// (~position).  range[i] = first and last sorted position below the node.
__global__ void k_lbvh_tree(const unsigned *keys, int n, int2 *children, int2 *range, int *parent_inner,
		int *parent_leaf)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n - 1)
		return;
	const int d = prefix_len(keys, n, i, i + 1) > prefix_len(keys, n, i, i - 1) ? 1 : -1;
	const int min_prefix = prefix_len(keys, n, i, i - d);
	int lmax = 2;
	while (prefix_len(keys, n, i, i + lmax * d) > min_prefix)
		lmax *= 2;
	int l = 0;
	for (int t = lmax / 2; t >= 1; t /= 2)
		if (prefix_len(keys, n, i, i + (l + t) * d) > min_prefix)
			l += t;
	const int j = i + l * d;
	const int node_prefix = prefix_len(keys, n, i, j);
	int s = 0;
	for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
		if (prefix_len(keys, n, i, i + (s + t) * d) > node_prefix)
			s += t;
		if (t == 1)
			break;
	}
	const int split = i + s * d + (d < 0 ? -1 : 0);
	const int lo = i < j ? i : j, hi = i < j ? j : i;
	const int left = split == lo ? ~split : split;
	const int right = split + 1 == hi ? ~(split + 1) : split + 1;
	children[i] = make_int2(left, right);
	range[i] = make_int2(lo, hi);
	if (left >= 0) parent_inner[left] = i; else parent_leaf[~left] = i;
	if (right >= 0) parent_inner[right] = i; else parent_leaf[~right] = i;
	if (i == 0)
		parent_inner[0] = -1;
}

struct Box6 { float lo[3], hi[3]; };

DEV Box6 triangle_box(const float *pos, unsigned orig)
{
	const float *p = pos + 9 * (size_t)orig;
	Box6 b;
	for (int a = 0; a < 3; a++) {
		b.lo[a] = __builtin_fminf(p[a], __builtin_fminf(p[3 + a], p[6 + a]));
		b.hi[a] = __builtin_fmaxf(p[a], __builtin_fmaxf(p[3 + a], p[6 + a]));
	}
	return b;
}

// One thread per sorted triangle climbs towards the root; at every inner node the
// first arrival stops and the second, which finds both children's boxes finished,
// computes the node's box.
__global__ void k_lbvh_fit(const float *pos, const unsigned *index, int n, const int2 *children,
		const int *parent_inner, const int *parent_leaf, unsigned *arrived, Box6 *boxes)
{
	const int leaf = blockIdx.x * blockDim.x + threadIdx.x;
	if (leaf >= n || n < 2)
		return;
	int node = parent_leaf[leaf];
	while (node >= 0) {
		__threadfence();
		if (atomicAdd(&arrived[node], 1u) == 0)
			return;
		__threadfence();
		const int2 c = children[node];
		const Box6 a = c.x >= 0 ? boxes[c.x] : triangle_box(pos, index[~c.x]);
		const Box6 b = c.y >= 0 ? boxes[c.y] : triangle_box(pos, index[~c.y]);
		Box6 u;
		for (int k = 0; k < 3; k++) {
			u.lo[k] = __builtin_fminf(a.lo[k], b.lo[k]);
			u.hi[k] = __builtin_fmaxf(a.hi[k], b.hi[k]);
		}
		boxes[node] = u;
		node = parent_inner[node];
	}
}

// child reference in the traversal layout: a subtree of at most RT_MAX_LEAF sorted
// triangles is a leaf (first position << 2 | count - 1, complemented)
DEV int emit_child(int child, const int2 *range)
{
	if (child < 0)
		return ~(((~child) << 2) | 0);
	const int2 r = range[child];
	const int count = r.y - r.x + 1;
	if (count <= RT_MAX_LEAF)
		return ~((r.x << 2) | (count - 1));
	return child;
}

__global__ void k_lbvh_emit(const float *pos, const unsigned *index, int n, const int2 *children,
		const int2 *range, const Box6 *boxes, DevBvhNode *nodes)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n - 1)
		return;
	const int2 c = children[i];
	const Box6 a = c.x >= 0 ? boxes[c.x] : triangle_box(pos, index[~c.x]);
	const Box6 b = c.y >= 0 ? boxes[c.y] : triangle_box(pos, index[~c.y]);
	DevBvhNode nd;
	nd.q[0] = a.lo[0]; nd.q[1] = a.lo[1]; nd.q[2] = a.lo[2]; nd.q[3] = a.hi[0];
	nd.q[4] = a.hi[1]; nd.q[5] = a.hi[2]; nd.q[6] = b.lo[0]; nd.q[7] = b.lo[1];
	nd.q[8] = b.lo[2]; nd.q[9] = b.hi[0]; nd.q[10] = b.hi[1]; nd.q[11] = b.hi[2];
	nd.q[12] = __builtin_bit_cast(float, emit_child(c.x, range));
	nd.q[13] = __builtin_bit_cast(float, emit_child(c.y, range));
	nd.q[14] = 0.0f; nd.q[15] = 0.0f;
	nodes[i] = nd;
}

// Triangles, corner normals and edge records in sorted order (the formats of
// scene_dev.h; the edges are the reference's vec3_sub(v, u), vec3_sub(w, u) in
// double, ray.c:334-335).
__global__ void k_lbvh_gather(const float *pos, const float *nrm, const unsigned *index, int n,
		DevTriangle *tris, DevTriangle *normals, DevTriEdges *edges)
{
	const int j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= n)
		return;
	const unsigned orig = index[j];
	const float *p = pos + 9 * (size_t)orig, *q = nrm + 9 * (size_t)orig;
	DevTriangle tp, tn;
	for (int c = 0; c < 3; c++) {
		for (int a = 0; a < 3; a++) {
			tp.v[4 * c + a] = p[3 * c + a];
			tn.v[4 * c + a] = q[3 * c + a];
		}
		tp.v[4 * c + 3] = 0.0f;
		tn.v[4 * c + 3] = 0.0f;
	}
	tp.v[3] = __builtin_bit_cast(float, orig);
	tris[j] = tp;
	normals[j] = tn;
	DevTriEdges te;
	for (int a = 0; a < 3; a++) {
		const double u = (double)p[a];
		te.u[a] = u;
		te.e1[a] = (double)p[3 + a] - u;
		te.e2[a] = (double)p[6 + a] - u;
	}
	te.orig = (int32_t)orig;
	te.pad[0] = 0;
	edges[j] = te;
}

__global__ void k_lbvh_depth(int n, const int *parent_leaf, const int *parent_inner, int *max_depth)
{
	const int leaf = blockIdx.x * blockDim.x + threadIdx.x;
	if (leaf >= n || n < 2)
		return;
	int depth = 1;
	for (int node = parent_leaf[leaf]; node >= 0; node = parent_inner[node])
		depth++;
	atomicMax(max_depth, depth);
}

struct Scratch {
	void *ptrs[16];
	int count = 0;
	template <typename T> hipError_t get(T **p, size_t n)
	{
		void *v = nullptr;
		hipError_t e = hipMalloc(&v, (n ? n : 1) * sizeof(T));
		if (e == hipSuccess)
			ptrs[count++] = v;
		*p = static_cast<T *>(v);
		return e;
	}
	~Scratch()
	{
		for (int i = 0; i < count; i++)
			(void)hipFree(ptrs[i]);
	}
};

#define TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return e_; } while (0)

} // namespace

hipError_t bvh_build_device(const float *d_pos, const float *d_nrm, int n, DevBvhNode *d_nodes,
		DevTriangle *d_tris, DevTriangle *d_normals, DevTriEdges *d_edges, int *root, int *max_depth,
		hipStream_t stream)
{
	Scratch s;
	unsigned *keys[2], *index[2], *counts, *bounds, *arrived;
	int2 *children, *range;
	int *parent_inner, *parent_leaf, *depth;
	Box6 *boxes;
	// elements per sort block: at least 2048, a multiple of 64, at most SORT_BLOCKS_MAX blocks
	int chunk = ((n + SORT_BLOCKS_MAX - 1) / SORT_BLOCKS_MAX + 63) / 64 * 64;
	if (chunk < 2048)
		chunk = 2048;
	const int sort_blocks = (n + chunk - 1) / chunk;
	for (int k = 0; k < 2; k++) {
		TRY(s.get(&keys[k], n));
		TRY(s.get(&index[k], n));
	}
	TRY(s.get(&counts, (size_t)RADIX * sort_blocks));
	TRY(s.get(&bounds, 8));
	TRY(s.get(&arrived, n));
	TRY(s.get(&children, n));
	TRY(s.get(&range, n));
	TRY(s.get(&parent_inner, n));
	TRY(s.get(&parent_leaf, n));
	TRY(s.get(&boxes, n));
	TRY(s.get(&depth, 1));

	const int tb = 256, blocks = (n + tb - 1) / tb;
	TRY(hipMemsetAsync(bounds, 0xff, 8 * sizeof(unsigned), stream));
	TRY(hipMemsetAsync(arrived, 0, (size_t)n * sizeof(unsigned), stream));
	TRY(hipMemsetAsync(depth, 0, sizeof(int), stream));
	hipLaunchKernelGGL(k_lbvh_bounds, dim3(blocks > 1024 ? 1024 : blocks), dim3(tb), 0, stream, d_pos, n, bounds);
	hipLaunchKernelGGL(k_lbvh_codes, dim3(blocks), dim3(tb), 0, stream, d_pos, n, bounds, keys[0], index[0]);
	int cur = 0;
	for (int shift = 0; shift < 32; shift += RADIX_BITS) {
		hipLaunchKernelGGL(k_radix_count, dim3(sort_blocks), dim3(64), 0, stream, keys[cur], n, chunk, shift, counts);
		hipLaunchKernelGGL(k_radix_scan, dim3(1), dim3(1024), 0, stream, counts, RADIX * sort_blocks);
		hipLaunchKernelGGL(k_radix_scatter, dim3(sort_blocks), dim3(64), 0, stream, keys[cur], index[cur], n, chunk,
				shift, counts, keys[cur ^ 1], index[cur ^ 1]);
		cur ^= 1;
	}
	if (n >= 2) {
		hipLaunchKernelGGL(k_lbvh_tree, dim3(blocks), dim3(tb), 0, stream, keys[cur], n, children, range,
				parent_inner, parent_leaf);
		hipLaunchKernelGGL(k_lbvh_fit, dim3(blocks), dim3(tb), 0, stream, d_pos, index[cur], n, children,
				parent_inner, parent_leaf, arrived, boxes);
		hipLaunchKernelGGL(k_lbvh_emit, dim3(blocks), dim3(tb), 0, stream, d_pos, index[cur], n, children, range,
				boxes, d_nodes);
		hipLaunchKernelGGL(k_lbvh_depth, dim3(blocks), dim3(tb), 0, stream, n, parent_leaf, parent_inner, depth);
	}
	hipLaunchKernelGGL(k_lbvh_gather, dim3(blocks), dim3(tb), 0, stream, d_pos, d_nrm, index[cur], n, d_tris,
			d_normals, d_edges);
	TRY(hipGetLastError());
	int h_depth = 0;
	TRY(hipMemcpyAsync(&h_depth, depth, sizeof(int), hipMemcpyDeviceToHost, stream));
	TRY(hipStreamSynchronize(stream));
	*max_depth = h_depth;
	// the whole mesh in one leaf, or the radix tree's root
	*root = n <= RT_MAX_LEAF ? ~((0 << 2) | (n - 1)) : 0;
	return hipSuccess;
}
