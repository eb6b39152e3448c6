/* Host-side BVH construction for one mesh. Internal. */
#ifndef RTAMD_BVH_BUILD_H
#define RTAMD_BVH_BUILD_H

#include <vector>
#include "scene_dev.h"

struct BvhResult {
	std::vector<DevBvhNode> nodes;   /* child indices relative to nodes[0] */
	std::vector<int> order;          /* leaf order -> triangle index in the mesh */
	int root;                        /* child reference of the root */
	int max_depth;
};

/* tri_pos: 9 floats per triangle (v0, v1, v2).  Returns false when the mesh is
 * empty or no tree shallow enough for the traversal stack could be built. */
bool bvh_build_mesh(const float *tri_pos, int ntri, BvhResult &out);

#endif
