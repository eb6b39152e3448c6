// Host-side BVH construction for one mesh (binned surface-area heuristic,
// leaves of at most four triangles, depth bounded by the traversal stack).
//
// This structure takes the place of the reference's kd-tree (mesh.c:350-451).
// That tree's shape depends on uninitialised memory (mesh.c:320) and is not
// observable in the output; only its nearest-hit result is reproduced, by the
// traversal in kernels.hip.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "bvh_build.h"

namespace {

struct Box {
	float lo[3], hi[3];
	void reset()
	{
		for (int a = 0; a < 3; a++) {
			lo[a] = INFINITY;
			hi[a] = -INFINITY;
		}
	}
	void grow(const Box &b)
	{
		for (int a = 0; a < 3; a++) {
			lo[a] = std::min(lo[a], b.lo[a]);
			hi[a] = std::max(hi[a], b.hi[a]);
		}
	}
	void grow(const float *p)
	{
		for (int a = 0; a < 3; a++) {
			lo[a] = std::min(lo[a], p[a]);
			hi[a] = std::max(hi[a], p[a]);
		}
	}
	float area() const
	{
		float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
		if (!(dx >= 0) || !(dy >= 0) || !(dz >= 0))
			return 0.0f;
		return 2.0f * (dx * dy + dy * dz + dz * dx);
	}
};

struct Builder {
	const float *pos;          // 9 floats per triangle
	int ntri;
	std::vector<Box> tbox;
	std::vector<float> centroid;   // 3 per triangle
	std::vector<int> order;
	std::vector<DevBvhNode> nodes;
	int max_depth = 0;
	bool sah;

#ifndef RT_MAX_LEAF
#define RT_MAX_LEAF 2
#endif
	static int leaf_ref(int first, int count) { return ~((first << 2) | (count - 1)); }

	// Returns a child reference (node index, or leaf encoding).
	int build(int first, int count, int depth)
	{
		max_depth = std::max(max_depth, depth);
		if (count <= RT_MAX_LEAF)
			return leaf_ref(first, count);

		Box cb;
		cb.reset();
		for (int i = first; i < first + count; i++)
			cb.grow(&centroid[3 * order[i]]);

		int mid = -1;
		if (sah && depth < 20) {
			const int NB = 16;
			float best_cost = INFINITY;
			int best_axis = -1, best_bin = -1;
			for (int axis = 0; axis < 3; axis++) {
				float ext = cb.hi[axis] - cb.lo[axis];
				if (!(ext > 0))
					continue;
				Box bins[NB];
				int cnt[NB];
				for (int b = 0; b < NB; b++) {
					bins[b].reset();
					cnt[b] = 0;
				}
				float scale = NB / ext;
				for (int i = first; i < first + count; i++) {
					int t = order[i];
					int b = std::min(NB - 1, std::max(0, (int)((centroid[3 * t + axis] - cb.lo[axis]) * scale)));
					bins[b].grow(tbox[t]);
					cnt[b]++;
				}
				float right_area[NB];
				int right_cnt[NB];
				Box acc;
				acc.reset();
				int c = 0;
				for (int b = NB - 1; b > 0; b--) {
					acc.grow(bins[b]);
					c += cnt[b];
					right_area[b] = acc.area();
					right_cnt[b] = c;
				}
				acc.reset();
				c = 0;
				for (int b = 0; b < NB - 1; b++) {
					acc.grow(bins[b]);
					c += cnt[b];
					if (c == 0 || right_cnt[b + 1] == 0)
						continue;
					float cost = acc.area() * c + right_area[b + 1] * right_cnt[b + 1];
					if (cost < best_cost) {
						best_cost = cost;
						best_axis = axis;
						best_bin = b;
					}
				}
			}
			if (best_axis >= 0) {
				float ext = cb.hi[best_axis] - cb.lo[best_axis];
				float scale = NB / ext;
				int *b = order.data() + first;
				int *m = std::partition(b, b + count, [&](int t) {
					int bin = std::min(NB - 1, std::max(0, (int)((centroid[3 * t + best_axis] - cb.lo[best_axis]) * scale)));
					return bin <= best_bin;
				});
				mid = (int)(m - b);
				if (mid <= 0 || mid >= count)
					mid = -1;
			}
		}
		if (mid < 0) {
			// median split along the widest centroid axis: always halves the set
			int axis = 0;
			float ext = cb.hi[0] - cb.lo[0];
			for (int a = 1; a < 3; a++)
				if (cb.hi[a] - cb.lo[a] > ext) {
					ext = cb.hi[a] - cb.lo[a];
					axis = a;
				}
			mid = count / 2;
			std::nth_element(order.begin() + first, order.begin() + first + mid, order.begin() + first + count,
					[&](int a, int b) {
						float ca = centroid[3 * a + axis], cb2 = centroid[3 * b + axis];
						return ca < cb2 || (ca == cb2 && a < b);
					});
		}

		int id = (int)nodes.size();
		nodes.emplace_back();
		Box b0, b1;
		b0.reset();
		b1.reset();
		for (int i = first; i < first + mid; i++)
			b0.grow(tbox[order[i]]);
		for (int i = first + mid; i < first + count; i++)
			b1.grow(tbox[order[i]]);
		int c0 = build(first, mid, depth + 1);
		int c1 = build(first + mid, count - mid, depth + 1);
		float *q = nodes[id].q;
		q[0] = b0.lo[0]; q[1] = b0.lo[1]; q[2] = b0.lo[2]; q[3] = b0.hi[0];
		q[4] = b0.hi[1]; q[5] = b0.hi[2]; q[6] = b1.lo[0]; q[7] = b1.lo[1];
		q[8] = b1.lo[2]; q[9] = b1.hi[0]; q[10] = b1.hi[1]; q[11] = b1.hi[2];
		std::memcpy(&q[12], &c0, 4);
		std::memcpy(&q[13], &c1, 4);
		q[14] = q[15] = 0.0f;
		return id;
	}
};

} // namespace

bool bvh_build_mesh(const float *tri_pos, int ntri, BvhResult &out)
{
	out.nodes.clear();
	out.order.clear();
	out.root = ~0;      // leaf(first 0, count 1): only used when ntri == 0 is rejected by the caller
	out.max_depth = 0;
	if (ntri <= 0)
		return false;
	for (int pass = 0; pass < 2; pass++) {
		Builder b;
		b.pos = tri_pos;
		b.ntri = ntri;
		b.sah = pass == 0;
		b.tbox.resize(ntri);
		b.centroid.resize(3 * (size_t)ntri);
		b.order.resize(ntri);
		for (int t = 0; t < ntri; t++) {
			b.tbox[t].reset();
			for (int c = 0; c < 3; c++)
				b.tbox[t].grow(tri_pos + 9 * (size_t)t + 3 * c);
			for (int a = 0; a < 3; a++)
				b.centroid[3 * (size_t)t + a] = 0.5f * (b.tbox[t].lo[a] + b.tbox[t].hi[a]);
			b.order[t] = t;
		}
		b.nodes.reserve((size_t)ntri);
		int root = b.build(0, ntri, 0);
		if (b.max_depth <= RT_MAX_STACK - 2 || pass == 1) {
			if (b.max_depth > RT_MAX_STACK - 2)
				return false;
			out.nodes = std::move(b.nodes);
			out.order = std::move(b.order);
			out.root = root;
			out.max_depth = b.max_depth;
			return true;
		}
	}
	return false;
}
