/* Minimal non-validating XML element tree: just enough for SDL scene files
 * (elements, attributes, comments, prolog, DOCTYPE).  Text nodes are dropped;
 * the reference only ever walks element children (scene.c:103,139,176,610). */
#ifndef RTAMD_XML_MINI_H
#define RTAMD_XML_MINI_H

typedef struct XmlAttr {
	char *name;
	char *value;
	struct XmlAttr *next;
} XmlAttr;

typedef struct XmlNode {
	char *name;
	XmlAttr *attrs;
	struct XmlNode *first_child;
	struct XmlNode *last_child;
	struct XmlNode *next;     /* next element sibling */
	struct XmlNode *parent;
} XmlNode;

/* Parses a whole document held in `text` (NUL-terminated).  Returns the root
 * element or NULL; on error `err` (size errlen) receives a message. */
XmlNode *xml_parse(const char *text, char *err, int errlen);
void xml_free(XmlNode *root);
/* Attribute value or NULL. */
const char *xml_attr(const XmlNode *node, const char *name);
int xml_child_count(const XmlNode *node);

#endif
