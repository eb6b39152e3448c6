/* Scene blob validation: every offset, count and index is checked against the
 * blob size before anything (host or device) walks the arrays. */
#include <string.h>

#include "front.h"

static int span_ok(uint64_t off, uint64_t bytes, uint64_t total)
{
	return off >= sizeof(rtamd_blob_header) && off <= total && bytes <= total - off &&
			(off & 7u) == 0;
}

int rtamd_blob_validate(const void *blob, size_t blob_size)
{
	if (!blob || blob_size < sizeof(rtamd_blob_header)) {
		rtamd_set_error("scene blob: too small");
		return RTAMD_ERR_ARG;
	}
	rtamd_blob_header h;
	memcpy(&h, blob, sizeof h);
	const uint8_t *base = (const uint8_t *)blob;
	if (h.magic != RTAMD_BLOB_MAGIC || h.total_size != blob_size) {
		rtamd_set_error("scene blob: bad magic or size");
		return RTAMD_ERR_ARG;
	}
	if (h.width <= 0 || h.height <= 0 || h.width > 65536 || h.height > 65536) {
		rtamd_set_error("scene blob: bad image size %dx%d", h.width, h.height);
		return RTAMD_ERR_ARG;
	}
	if (h.antialiasing && (h.aa_samples <= 0 || h.aa_samples > 64)) {
		rtamd_set_error("scene blob: aa_samples %d out of range 1..64", h.aa_samples);
		return RTAMD_ERR_ARG;
	}
	if (h.shadow_samples < 0 || h.shadow_samples > 64 || h.reflection_samples < 0 ||
			h.reflection_samples > 256 || h.max_reflections > 64) {
		rtamd_set_error("scene blob: sample counts out of range");
		return RTAMD_ERR_ARG;
	}
	if (h.num_lights < 0 || h.num_lights > RTAMD_MAX_LIGHTS || h.num_materials < 0 ||
			h.num_surfaces < 0 || h.num_meshes < 0) {
		rtamd_set_error("scene blob: bad counts");
		return RTAMD_ERR_ARG;
	}
	if (!span_ok(h.off_lights, sizeof(rtamd_light) * (uint64_t)h.num_lights, blob_size) ||
			!span_ok(h.off_materials, sizeof(rtamd_material) * (uint64_t)h.num_materials, blob_size) ||
			!span_ok(h.off_surfaces, sizeof(rtamd_surface) * (uint64_t)h.num_surfaces, blob_size) ||
			!span_ok(h.off_meshes, sizeof(rtamd_mesh) * (uint64_t)h.num_meshes, blob_size)) {
		rtamd_set_error("scene blob: table offset out of bounds");
		return RTAMD_ERR_ARG;
	}
	const rtamd_mesh *meshes = (const rtamd_mesh *)(base + h.off_meshes);
	for (int m = 0; m < h.num_meshes; m++) {
		const rtamd_mesh *d = &meshes[m];
		if (d->num_vertices < 0 || d->num_normals < 0 || d->num_triangles < 0 ||
				!span_ok(d->off_vertex, 24ull * (uint64_t)d->num_vertices, blob_size) ||
				(d->num_normals &&
				 !span_ok(d->off_normal, 24ull * (uint64_t)d->num_normals, blob_size)) ||
				!span_ok(d->off_triangle, 24ull * (uint64_t)d->num_triangles, blob_size)) {
			rtamd_set_error("scene blob: mesh %d arrays out of bounds", m);
			return RTAMD_ERR_ARG;
		}
		if (d->off_name && (d->off_name >= blob_size ||
				!memchr(base + d->off_name, 0, blob_size - d->off_name))) {
			rtamd_set_error("scene blob: mesh %d name is not terminated inside the blob", m);
			return RTAMD_ERR_ARG;
		}
		const int32_t *tri = (const int32_t *)(base + d->off_triangle);
		for (int64_t t = 0; t < d->num_triangles; t++)
			for (int c = 0; c < 3; c++) {
				int32_t v = tri[6 * t + c], n = tri[6 * t + 3 + c];
				if (v < 0 || v >= d->num_vertices || n < 0 || n >= d->num_normals) {
					rtamd_set_error("scene blob: mesh %d triangle %lld has a vertex or "
							"normal index out of range (a mesh needs vn normals)",
							m, (long long)t);
					return RTAMD_ERR_ARG;
				}
			}
	}
	const rtamd_surface *surfs = (const rtamd_surface *)(base + h.off_surfaces);
	for (int s = 0; s < h.num_surfaces; s++) {
		const rtamd_surface *sf = &surfs[s];
		if (sf->shape_type < RTAMD_SHAPE_PLANE || sf->shape_type > RTAMD_SHAPE_MESH ||
				sf->material < 0 || sf->material >= h.num_materials ||
				(sf->shape_type == RTAMD_SHAPE_MESH &&
				 (sf->mesh < 0 || sf->mesh >= h.num_meshes))) {
			rtamd_set_error("scene blob: surface %d is malformed", s);
			return RTAMD_ERR_ARG;
		}
	}
	if (h.has_cubemap) {
		if (!span_ok(h.off_cubemap, sizeof(rtamd_cubemap), blob_size)) {
			rtamd_set_error("scene blob: cubemap header out of bounds");
			return RTAMD_ERR_ARG;
		}
		const rtamd_cubemap *c = (const rtamd_cubemap *)(base + h.off_cubemap);
		for (int f = 0; f < 6; f++)
			if (c->width[f] <= 0 || c->height[f] <= 0 || c->width[f] > 16384 ||
					c->height[f] > 16384 ||
					!span_ok(c->off_face[f], 16ull * (uint64_t)c->width[f] * (uint64_t)c->height[f],
						blob_size)) {
				rtamd_set_error("scene blob: cubemap face %d out of bounds", f);
				return RTAMD_ERR_ARG;
			}
	}
	return 0;
}
