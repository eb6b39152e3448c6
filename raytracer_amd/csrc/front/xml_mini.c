#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "xml_mini.h"

typedef struct Parser {
	const char *p;
	int line;
	char *err;
	int errlen;
	int failed;
} Parser;

static void fail(Parser *ps, const char *msg)
{
	if (!ps->failed && ps->err)
		snprintf(ps->err, ps->errlen, "XML line %d: %s", ps->line, msg);
	ps->failed = 1;
}

static void advance(Parser *ps, int n)
{
	while (n-- > 0 && *ps->p) {
		if (*ps->p == '\n')
			ps->line++;
		ps->p++;
	}
}

static void skip_ws(Parser *ps)
{
	while (*ps->p && isspace((unsigned char)*ps->p))
		advance(ps, 1);
}

static int starts(Parser *ps, const char *s)
{
	return strncmp(ps->p, s, strlen(s)) == 0;
}

/* Skips until after the first occurrence of `end`. */
static void skip_past(Parser *ps, const char *end)
{
	const char *q = strstr(ps->p, end);
	if (!q) {
		fail(ps, "unterminated construct");
		return;
	}
	advance(ps, (int)(q - ps->p) + (int)strlen(end));
}

static void skip_doctype(Parser *ps)
{
	/* <!DOCTYPE name SYSTEM "x" [ internal subset ]> */
	int depth = 0;
	while (*ps->p) {
		char c = *ps->p;
		if (c == '[')
			depth++;
		else if (c == ']')
			depth--;
		else if (c == '>' && depth <= 0) {
			advance(ps, 1);
			return;
		}
		advance(ps, 1);
	}
	fail(ps, "unterminated DOCTYPE");
}

static int is_name_char(int c)
{
	return isalnum(c) || c == '_' || c == '-' || c == '.' || c == ':';
}

static char *parse_name(Parser *ps)
{
	const char *s = ps->p;
	while (is_name_char((unsigned char)*ps->p))
		advance(ps, 1);
	if (ps->p == s) {
		fail(ps, "expected a name");
		return NULL;
	}
	size_t n = (size_t)(ps->p - s);
	char *out = malloc(n + 1);
	memcpy(out, s, n);
	out[n] = '\0';
	return out;
}

/* Attribute value with the five predefined entities and numeric character
 * references decoded; literal tab/newline become spaces (XML 1.0 3.3.3). */
static char *parse_attr_value(Parser *ps)
{
	char quote = *ps->p;
	if (quote != '"' && quote != '\'') {
		fail(ps, "expected a quoted attribute value");
		return NULL;
	}
	advance(ps, 1);
	size_t cap = 32, len = 0;
	char *out = malloc(cap);
	while (*ps->p && *ps->p != quote) {
		char c = *ps->p;
		if (c == '&') {
			static const struct { const char *ent; char ch; } ents[] = {
				{"&amp;", '&'}, {"&lt;", '<'}, {"&gt;", '>'},
				{"&quot;", '"'}, {"&apos;", '\''} };
			int done = 0;
			for (unsigned i = 0; i < sizeof ents / sizeof ents[0]; i++)
				if (starts(ps, ents[i].ent)) {
					c = ents[i].ch;
					advance(ps, (int)strlen(ents[i].ent) - 1);
					done = 1;
					break;
				}
			if (!done && ps->p[1] == '#') {
				char *end;
				long v = (ps->p[2] == 'x') ? strtol(ps->p + 3, &end, 16)
				                            : strtol(ps->p + 2, &end, 10);
				if (*end == ';' && v > 0 && v < 128) {
					c = (char)v;
					advance(ps, (int)(end - ps->p));
				}
			}
		} else if (c == '\t' || c == '\n' || c == '\r')
			c = ' ';
		if (len + 2 > cap) {
			cap *= 2;
			out = realloc(out, cap);
		}
		out[len++] = c;
		advance(ps, 1);
	}
	if (*ps->p != quote) {
		fail(ps, "unterminated attribute value");
		free(out);
		return NULL;
	}
	advance(ps, 1);
	out[len] = '\0';
	return out;
}

static void skip_misc(Parser *ps)
{
	for (;;) {
		skip_ws(ps);
		if (starts(ps, "<!--"))
			skip_past(ps, "-->");
		else if (starts(ps, "<?"))
			skip_past(ps, "?>");
		else if (starts(ps, "<!DOCTYPE"))
			skip_doctype(ps);
		else
			return;
		if (ps->failed)
			return;
	}
}

static XmlNode *parse_element(Parser *ps, XmlNode *parent)
{
	if (*ps->p != '<') {
		fail(ps, "expected '<'");
		return NULL;
	}
	advance(ps, 1);
	XmlNode *node = calloc(1, sizeof *node);
	node->parent = parent;
	node->name = parse_name(ps);
	if (ps->failed)
		return node;

	XmlAttr *tail = NULL;
	for (;;) {
		skip_ws(ps);
		if (*ps->p == '/' && ps->p[1] == '>') {
			advance(ps, 2);
			return node;
		}
		if (*ps->p == '>') {
			advance(ps, 1);
			break;
		}
		XmlAttr *a = calloc(1, sizeof *a);
		a->name = parse_name(ps);
		if (tail)
			tail->next = a;
		else
			node->attrs = a;
		tail = a;
		if (ps->failed)
			return node;
		skip_ws(ps);
		if (*ps->p != '=') {
			fail(ps, "expected '=' after attribute name");
			return node;
		}
		advance(ps, 1);
		skip_ws(ps);
		a->value = parse_attr_value(ps);
		if (ps->failed)
			return node;
	}

	/* content */
	for (;;) {
		while (*ps->p && *ps->p != '<')
			advance(ps, 1);          /* character data: ignored */
		if (!*ps->p) {
			fail(ps, "unexpected end of document inside element");
			return node;
		}
		if (starts(ps, "<!--")) {
			skip_past(ps, "-->");
		} else if (starts(ps, "<![CDATA[")) {
			skip_past(ps, "]]>");
		} else if (starts(ps, "<?")) {
			skip_past(ps, "?>");
		} else if (starts(ps, "</")) {
			advance(ps, 2);
			char *close = parse_name(ps);
			if (close && strcmp(close, node->name) != 0)
				fail(ps, "mismatched closing tag");
			free(close);
			skip_ws(ps);
			if (*ps->p != '>')
				fail(ps, "expected '>' in closing tag");
			else
				advance(ps, 1);
			return node;
		} else {
			XmlNode *child = parse_element(ps, node);
			if (child) {
				if (node->last_child)
					node->last_child->next = child;
				else
					node->first_child = child;
				node->last_child = child;
			}
		}
		if (ps->failed)
			return node;
	}
}

XmlNode *xml_parse(const char *text, char *err, int errlen)
{
	Parser ps = { text, 1, err, errlen, 0 };
	if (err && errlen > 0)
		err[0] = '\0';
	/* optional UTF-8 byte order mark */
	if ((unsigned char)ps.p[0] == 0xEF && (unsigned char)ps.p[1] == 0xBB &&
			(unsigned char)ps.p[2] == 0xBF)
		ps.p += 3;
	skip_misc(&ps);
	if (ps.failed)
		return NULL;
	XmlNode *root = parse_element(&ps, NULL);
	if (ps.failed) {
		xml_free(root);
		return NULL;
	}
	return root;
}

void xml_free(XmlNode *node)
{
	if (!node)
		return;
	XmlNode *c = node->first_child;
	while (c) {
		XmlNode *n = c->next;
		xml_free(c);
		c = n;
	}
	XmlAttr *a = node->attrs;
	while (a) {
		XmlAttr *n = a->next;
		free(a->name);
		free(a->value);
		free(a);
		a = n;
	}
	free(node->name);
	free(node);
}

const char *xml_attr(const XmlNode *node, const char *name)
{
	for (const XmlAttr *a = node->attrs; a; a = a->next)
		if (a->name && strcmp(a->name, name) == 0)
			return a->value;
	return NULL;
}

int xml_child_count(const XmlNode *node)
{
	int n = 0;
	for (const XmlNode *c = node->first_child; c; c = c->next)
		n++;
	return n;
}
