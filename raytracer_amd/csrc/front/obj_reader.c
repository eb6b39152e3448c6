/* Wavefront OBJ reader for triangle meshes.
 *
 * Produces what mesh_load (mesh.c:190-231) builds through the objreader
 * callbacks (objreader/usercallbacks.h:22-38): every coordinate goes through
 * (float)atof (objreader/wavefrontobj.l:69-72, or (float)atoi for integer
 * tokens, wavefrontobj.y:97-99) before it is widened into the double-valued
 * vertex/normal arrays (mesh.c:93-95,120-122); face corners are v, v/vt, v//vn
 * or v/vt/vn with 1-based indices, indices below 1 are an error
 * (wavefrontobj.y:215-248), and the stored index is value-1 (mesh.c:149-151).
 * Only triangles are accepted: the reference's consumer has room for three
 * corners per face (mesh.h:8-12).
 */
#include <ctype.h>
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "front.h"

void front_mesh_free(FrontMesh *mesh)
{
	if (!mesh)
		return;
	free(mesh->vertex);
	free(mesh->normal);
	free(mesh->triangle);
	memset(mesh, 0, sizeof *mesh);
}

static const char *skip_blank(const char *p)
{
	while (*p == ' ' || *p == '\t')
		p++;
	return p;
}

/* One coordinate token narrowed to float the way the lexer does it. */
static int parse_coord(const char **pp, float *out)
{
	const char *p = skip_blank(*pp);
	char *end;
	double d = strtod(p, &end);
	if (end == p)
		return 0;
	*out = (float)d;
	*pp = end;
	return 1;
}

static int parse_corner(const char **pp, long idx[3])
{
	const char *p = skip_blank(*pp);
	char *end;
	idx[0] = idx[1] = idx[2] = 0;
	if (!isdigit((unsigned char)*p) && *p != '-' && *p != '+')
		return 0;
	idx[0] = strtol(p, &end, 10);
	if (end == p)
		return 0;
	p = end;
	if (*p == '/') {
		p++;
		if (*p == '/') {
			p++;
			idx[2] = strtol(p, &end, 10);
			if (end == p)
				return -1;
			p = end;
			if (idx[2] < 1)
				return -1;
		} else {
			idx[1] = strtol(p, &end, 10);
			if (end == p)
				return -1;
			p = end;
			if (idx[1] < 1)
				return -1;
			if (*p == '/') {
				p++;
				idx[2] = strtol(p, &end, 10);
				if (end == p)
					return -1;
				p = end;
				if (idx[2] < 1)
					return -1;
			}
		}
	}
	if (idx[0] < 1)
		return -1;
	*pp = p;
	return 1;
}

static int read_all(const char *path, char **text)
{
	FILE *f = fopen(path, "rb");
	if (!f) {
		rtamd_set_error("Opening file %s failed: %s", path, strerror(errno));
		return RTAMD_ERR_IO;
	}
	fseek(f, 0, SEEK_END);
	long n = ftell(f);
	fseek(f, 0, SEEK_SET);
	char *buf = malloc((size_t)n + 2);
	if (!buf) {
		fclose(f);
		return RTAMD_ERR_NOMEM;
	}
	size_t got = fread(buf, 1, (size_t)n, f);
	fclose(f);
	buf[got] = '\n';
	buf[got + 1] = '\0';
	*text = buf;
	return 0;
}

int front_obj_load(const char *path, FrontMesh *mesh)
{
	char *text = NULL;
	int rc = read_all(path, &text);
	if (rc)
		return rc;
	memset(mesh, 0, sizeof *mesh);

	/* Pass 1: count (mesh.c:47-73). */
	int nv = 0, nn = 0, nt = 0, nf = 0;
	for (const char *p = text; *p;) {
		const char *q = skip_blank(p);
		if (q[0] == 'v' && (q[1] == ' ' || q[1] == '\t'))
			nv++;
		else if (q[0] == 'v' && q[1] == 'n' && (q[2] == ' ' || q[2] == '\t'))
			nn++;
		else if (q[0] == 'v' && q[1] == 't' && (q[2] == ' ' || q[2] == '\t'))
			nt++;
		else if (q[0] == 'f' && (q[1] == ' ' || q[1] == '\t'))
			nf++;
		while (*p && *p != '\n')
			p++;
		if (*p)
			p++;
	}
	mesh->num_vertices = nv;
	mesh->num_normals = nn;
	mesh->num_texcoords = nt;
	mesh->num_triangles = nf;
	mesh->vertex = calloc((size_t)(nv ? nv : 1) * 3, sizeof(double));
	mesh->normal = nn ? calloc((size_t)nn * 3, sizeof(double)) : NULL;
	mesh->triangle = calloc((size_t)(nf ? nf : 1) * 6, sizeof(int32_t));
	if (!mesh->vertex || !mesh->triangle || (nn && !mesh->normal)) {
		free(text);
		front_mesh_free(mesh);
		return RTAMD_ERR_NOMEM;
	}

	/* Pass 2: fill (mesh.c:158-188). */
	int iv = 0, in = 0, it = 0, line = 0;
	for (char *p = text; *p;) {
		char *eol = p;
		while (*eol && *eol != '\n')
			eol++;
		char saved = *eol;
		*eol = '\0';
		line++;
		/* strip a trailing carriage return */
		if (eol > p && eol[-1] == '\r')
			eol[-1] = ' ';
		const char *q = skip_blank(p);
		if (q[0] == 'v' && (q[1] == ' ' || q[1] == '\t')) {
			const char *c = q + 1;
			float f[3];
			if (!parse_coord(&c, &f[0]) || !parse_coord(&c, &f[1]) || !parse_coord(&c, &f[2]))
				goto syntax;
			mesh->vertex[3 * iv + 0] = f[0];
			mesh->vertex[3 * iv + 1] = f[1];
			mesh->vertex[3 * iv + 2] = f[2];
			iv++;
		} else if (q[0] == 'v' && q[1] == 'n' && (q[2] == ' ' || q[2] == '\t')) {
			const char *c = q + 2;
			float f[3];
			if (!parse_coord(&c, &f[0]) || !parse_coord(&c, &f[1]) || !parse_coord(&c, &f[2]))
				goto syntax;
			mesh->normal[3 * in + 0] = f[0];
			mesh->normal[3 * in + 1] = f[1];
			mesh->normal[3 * in + 2] = f[2];
			in++;
		} else if (q[0] == 'f' && (q[1] == ' ' || q[1] == '\t')) {
			const char *c = q + 1;
			long idx[3];
			int corners = 0;
			for (;;) {
				int r = parse_corner(&c, idx);
				if (r == 0)
					break;
				if (r < 0)
					goto syntax;
				if (corners >= 3) {
					rtamd_set_error("%s line %d: face with more than three "
							"corners (only triangle meshes are supported)",
							path, line);
					goto bad;
				}
				if (idx[0] > nv || idx[2] > nn) {
					rtamd_set_error("%s line %d: index out of range", path, line);
					goto bad;
				}
				mesh->triangle[6 * it + corners] = (int32_t)idx[0] - 1;
				mesh->triangle[6 * it + 3 + corners] = (int32_t)idx[2] - 1;
				corners++;
			}
			c = skip_blank(c);
			if (*c != '\0' || corners != 3)
				goto syntax;
			it++;
		}
		/* everything else (vt, g, o, s, usemtl, mtllib, comments) carries
		 * nothing the tracer reads */
		*eol = saved;
		p = *eol ? eol + 1 : eol;
		continue;
syntax:
		rtamd_set_error("Error parsing file %s (line %d)", path, line);
bad:
		free(text);
		front_mesh_free(mesh);
		return RTAMD_ERR_IO;
	}
	free(text);
	return 0;
}
