/* BVH construction on the GPU for one mesh (bvh_device.hip). Internal. */
#ifndef RTAMD_BVH_DEVICE_H
#define RTAMD_BVH_DEVICE_H

#include <hip/hip_runtime.h>
#include "scene_dev.h"

/* d_pos, d_nrm: DEVICE float[9 * n], the corner positions and corner normals of the
 * mesh's triangles in the mesh's own order.  Writes the mesh's part of the scene
 * arrays (DEVICE pointers already offset to the mesh's first node / triangle):
 * at most n - 1 nodes, n triangles, normals and edge records in leaf order.
 * *root = child reference of the root, *max_depth = levels from the root to the
 * deepest triangle (the caller compares it with the traversal stack).
 * Synchronises `stream` before it returns. */
hipError_t bvh_build_device(const float *d_pos, const float *d_nrm, int n, DevBvhNode *d_nodes,
		DevTriangle *d_tris, DevTriangle *d_normals, DevTriEdges *d_edges, int *root, int *max_depth,
		hipStream_t stream);

#endif
