/* yaml.c -- ls_hs_load_yaml_config / ls_hs_destroy_yaml_config in plain C.
 *
 * The reference loads its inputs through lattice-symmetries-haskell
 * (/root/reference/src/FFI.chpl:121-126,208-209; /root/reference/src/ForeignTypes.chpl:261-288: loadConfigFromYaml clones
 * `basis`, `hamiltonian` and the `observables` out of an ls_hs_yaml_config and destroys it).  That library is not part of
 * the reference tree, so this file implements the same entry points for the YAML subset its data files use
 * (the YAML files under /root/reference/data):
 *
 *   basis:        number_spins, hamming_weight (int | null), spin_inversion (1 | -1 | absent), particle (spin-1/2),
 *                 symmetries: [{permutation: [...], sector: int}]
 *   hamiltonian:  terms: [{expression: "0.8 × σˣ₀ σˣ₁", sites: [[i, j], ...]}]     (other keys hold YAML anchors)
 *   observables:  [ {terms: ...}, ... ]                                              (always empty in the reference's files)
 *
 * Syntax handled: block mappings and block sequences by indentation, flow sequences and mappings (nested, over several lines),
 * anchors (&name) and aliases (*name), double- and single-quoted and plain scalars, comments.
 * Every expression is a monomial of single-site operators; each maps a basis state to at most one basis state, so a
 * monomial on a tuple of sites compiles into a handful of non-branching terms (v, m, r, x, s) -- the same symbolic
 * compilation as the Python mirror (distributed-matvec_amd/config.py); merging, cancellation and the grouping by flip mask
 * happen in ls_hs_create_operator_from_terms.  Conventions: site i <-> bit i; bit 0 = spin up; S^a = sigma^a / 2.
 */
#define _GNU_SOURCE
#include <ctype.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/ls_hs.h"

int ls_amd_internal_error(char const *fmt, ...); /* host.c: formats into ls_amd_last_error(), returns -1 */
void ls_amd_internal_clear_error(void);

/* ------------------------------------------------------------------------------------------------ */
/* a small document tree                                                                            */
/* ------------------------------------------------------------------------------------------------ */
typedef enum { Y_SCALAR = 0, Y_MAP = 1, Y_SEQ = 2 } ykind;
typedef struct ynode {
    ykind kind;
    char *str;          /* scalar text (NULL for a null node) */
    int quoted;         /* the scalar was written in quotes: never null / a number */
    int n, cap;
    char **keys;        /* maps */
    struct ynode **vals;
} ynode;

static ynode *y_new(ykind k) {
    ynode *n = (ynode *)calloc(1, sizeof(*n));
    n->kind = k;
    return n;
}
static void y_free(ynode *n) {
    if (!n) return;
    for (int i = 0; i < n->n; ++i) {
        if (n->keys) free(n->keys[i]);
        y_free(n->vals[i]);
    }
    free(n->keys); free(n->vals); free(n->str); free(n);
}
static void y_push(ynode *n, char *key, ynode *val) {
    if (n->n == n->cap) {
        n->cap = n->cap ? 2 * n->cap : 8;
        n->vals = (ynode **)realloc(n->vals, sizeof(ynode *) * (size_t)n->cap);
        if (n->kind == Y_MAP) n->keys = (char **)realloc(n->keys, sizeof(char *) * (size_t)n->cap);
    }
    if (n->kind == Y_MAP) n->keys[n->n] = key;
    n->vals[n->n++] = val;
}
static ynode *y_copy(ynode const *a) {
    if (!a) return NULL;
    ynode *n = y_new(a->kind);
    n->quoted = a->quoted;
    if (a->str) n->str = strdup(a->str);
    for (int i = 0; i < a->n; ++i) y_push(n, a->keys ? strdup(a->keys[i]) : NULL, y_copy(a->vals[i]));
    return n;
}
static ynode *y_get(ynode const *map, char const *key) {
    if (!map || map->kind != Y_MAP) return NULL;
    for (int i = 0; i < map->n; ++i)
        if (strcmp(map->keys[i], key) == 0) return map->vals[i];
    return NULL;
}
static int y_is_null(ynode const *n) {
    if (!n) return 1;
    if (n->kind != Y_SCALAR) return 0;
    if (!n->str) return 1;
    if (n->quoted) return 0;
    return n->str[0] == 0 || strcmp(n->str, "null") == 0 || strcmp(n->str, "~") == 0 || strcmp(n->str, "Null") == 0 ||
           strcmp(n->str, "NULL") == 0;
}

/* ------------------------------------------------------------------------------------------------ */
/* parser                                                                                           */
/* ------------------------------------------------------------------------------------------------ */
typedef struct {
    char const *s;
    size_t pos, len;
    char err[256];
    int n_anchors;
    char *anchor_names[64];
    ynode *anchor_nodes[64]; /* borrowed: they live in the tree */
} yparser;

static int y_fail(yparser *p, char const *fmt, ...) {
    if (p->err[0]) return -1;
    int line = 1;
    for (size_t i = 0; i < p->pos && i < p->len; ++i) line += p->s[i] == '\n';
    int k = snprintf(p->err, sizeof(p->err), "line %d: ", line);
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(p->err + k, sizeof(p->err) - (size_t)k, fmt, ap);
    va_end(ap);
    return -1;
}
/* moves to the first character of the next line that holds content; returns its indentation, or -1 at the end */
static int y_next_line(yparser *p) {
    for (;;) {
        /* p->pos is at the start of a line here */
        size_t q = p->pos;
        int ind = 0;
        while (q < p->len && p->s[q] == ' ') { ++q; ++ind; }
        if (q >= p->len) { p->pos = q; return -1; }
        if (p->s[q] == '\n' || p->s[q] == '\r' || p->s[q] == '#' || (p->s[q] == '-' && q + 2 < p->len && p->s[q + 1] == '-' && p->s[q + 2] == '-')) {
            while (q < p->len && p->s[q] != '\n') ++q; /* blank line, comment, document marker */
            p->pos = q < p->len ? q + 1 : q;
            continue;
        }
        if (p->s[q] == '\t') { p->pos = q; y_fail(p, "tabs are not allowed in indentation"); return -1; }
        p->pos = q;
        return ind;
    }
}
static int y_peek_indent(yparser *p) {
    size_t const save = p->pos;
    int const ind = y_next_line(p);
    p->pos = save;
    return ind;
}
static void y_to_line_start(yparser *p) { /* after an inline value: drop the rest of the line (spaces, comment) */
    while (p->pos < p->len && p->s[p->pos] != '\n') ++p->pos;
    if (p->pos < p->len) ++p->pos;
}
static void y_skip_flow_ws(yparser *p) { /* inside [...]: spaces, newlines, comments */
    for (;;) {
        while (p->pos < p->len && isspace((unsigned char)p->s[p->pos])) ++p->pos;
        if (p->pos < p->len && p->s[p->pos] == '#') { while (p->pos < p->len && p->s[p->pos] != '\n') ++p->pos; continue; }
        return;
    }
}
static char *y_substr(char const *a, size_t n) {
    while (n > 0 && isspace((unsigned char)a[n - 1])) --n;
    char *r = (char *)malloc(n + 1);
    memcpy(r, a, n);
    r[n] = 0;
    return r;
}
static ynode *y_quoted(yparser *p) {
    char const q = p->s[p->pos++];
    size_t cap = 64, n = 0;
    char *buf = (char *)malloc(cap);
    while (p->pos < p->len && p->s[p->pos] != q) {
        char c = p->s[p->pos++];
        if (c == '\\' && q == '"' && p->pos < p->len) {
            char const e = p->s[p->pos++];
            c = e == 'n' ? '\n' : e == 't' ? '\t' : e;
        }
        if (n + 2 > cap) { cap *= 2; buf = (char *)realloc(buf, cap); }
        buf[n++] = c;
    }
    if (p->pos >= p->len) { free(buf); y_fail(p, "unterminated string"); return NULL; }
    ++p->pos;
    buf[n] = 0;
    ynode *nd = y_new(Y_SCALAR);
    nd->str = buf;
    nd->quoted = 1;
    return nd;
}
static ynode *y_flow(yparser *p);
static ynode *y_flow_map(yparser *p);
static ynode *y_flow_item(yparser *p) {
    y_skip_flow_ws(p);
    if (p->pos >= p->len) { y_fail(p, "unterminated flow collection"); return NULL; }
    char const c = p->s[p->pos];
    if (c == '[') return y_flow(p);
    if (c == '{') return y_flow_map(p);
    if (c == '"' || c == '\'') return y_quoted(p);
    size_t const a = p->pos;
    while (p->pos < p->len && p->s[p->pos] != ',' && p->s[p->pos] != ']' && p->s[p->pos] != '}' && p->s[p->pos] != '\n' && p->s[p->pos] != '#') ++p->pos;
    ynode *nd = y_new(Y_SCALAR);
    nd->str = y_substr(p->s + a, p->pos - a);
    return nd;
}
static ynode *y_flow(yparser *p) { /* p->pos at '[' */
    ++p->pos;
    ynode *seq = y_new(Y_SEQ);
    for (;;) {
        y_skip_flow_ws(p);
        if (p->pos >= p->len) { y_free(seq); y_fail(p, "unterminated flow sequence"); return NULL; }
        if (p->s[p->pos] == ']') { ++p->pos; return seq; }
        ynode *it = y_flow_item(p);
        if (!it) { y_free(seq); return NULL; }
        y_push(seq, NULL, it);
        y_skip_flow_ws(p);
        if (p->pos < p->len && p->s[p->pos] == ',') { ++p->pos; continue; }
        if (p->pos < p->len && p->s[p->pos] == ']') { ++p->pos; return seq; }
        y_free(seq);
        y_fail(p, "expected ',' or ']' in a flow sequence");
        return NULL;
    }
}
static ynode *y_flow_map(yparser *p) { /* p->pos at '{' */
    ++p->pos;
    ynode *map = y_new(Y_MAP);
    for (;;) {
        y_skip_flow_ws(p);
        if (p->pos >= p->len) { y_free(map); y_fail(p, "unterminated flow mapping"); return NULL; }
        if (p->s[p->pos] == '}') { ++p->pos; return map; }
        size_t const a = p->pos;
        while (p->pos < p->len && p->s[p->pos] != ':' && p->s[p->pos] != '}' && p->s[p->pos] != ',' && p->s[p->pos] != '\n') ++p->pos;
        if (p->pos >= p->len || p->s[p->pos] != ':') { y_free(map); y_fail(p, "expected 'key: value' in a flow mapping"); return NULL; }
        char *key = y_substr(p->s + a, p->pos - a);
        ++p->pos;
        ynode *val = y_flow_item(p);
        if (!val) { free(key); y_free(map); return NULL; }
        y_push(map, key, val);
        y_skip_flow_ws(p);
        if (p->pos < p->len && p->s[p->pos] == ',') { ++p->pos; continue; }
        if (p->pos < p->len && p->s[p->pos] == '}') { ++p->pos; return map; }
        y_free(map);
        y_fail(p, "expected ',' or '}' in a flow mapping");
        return NULL;
    }
}
/* value written on the rest of the current line (after "key: " or "- "); the cursor ends at the start of the next line */
static ynode *y_inline(yparser *p) {
    while (p->pos < p->len && p->s[p->pos] == ' ') ++p->pos;
    char *anchor = NULL;
    if (p->pos < p->len && p->s[p->pos] == '&') {
        size_t const a = ++p->pos;
        while (p->pos < p->len && !isspace((unsigned char)p->s[p->pos])) ++p->pos;
        anchor = y_substr(p->s + a, p->pos - a);
        while (p->pos < p->len && p->s[p->pos] == ' ') ++p->pos;
    }
    ynode *val = NULL;
    if (p->pos >= p->len || p->s[p->pos] == '\n' || p->s[p->pos] == '#') {
        val = y_new(Y_SCALAR); /* null */
    } else if (p->s[p->pos] == '*') {
        size_t const a = ++p->pos;
        while (p->pos < p->len && !isspace((unsigned char)p->s[p->pos])) ++p->pos;
        char *name = y_substr(p->s + a, p->pos - a);
        for (int i = p->n_anchors - 1; i >= 0 && !val; --i) /* (a name bound twice: the later binding, one copy) */
            if (strcmp(p->anchor_names[i], name) == 0) val = y_copy(p->anchor_nodes[i]);
        if (!val) y_fail(p, "unknown alias *%s", name);
        free(name);
    } else if (p->s[p->pos] == '[') {
        val = y_flow(p);
    } else if (p->s[p->pos] == '{') {
        val = y_flow_map(p);
    } else if (p->s[p->pos] == '"' || p->s[p->pos] == '\'') {
        val = y_quoted(p);
    } else {
        size_t const a = p->pos;
        while (p->pos < p->len && p->s[p->pos] != '\n' && !(p->s[p->pos] == '#' && p->pos > a && p->s[p->pos - 1] == ' ')) ++p->pos;
        val = y_new(Y_SCALAR);
        val->str = y_substr(p->s + a, p->pos - a);
    }
    if (val && anchor) {
        if (p->n_anchors < 64) { p->anchor_names[p->n_anchors] = anchor; p->anchor_nodes[p->n_anchors++] = val; anchor = NULL; }
        else y_fail(p, "too many anchors");
    }
    free(anchor);
    if (val) y_to_line_start(p);
    return val;
}
static ynode *y_block(yparser *p, int min_indent);
/* "&name" followed by the end of the line: the anchor belongs to the block collection on the following lines.  Returns the
 * name (malloc'ed) and moves *q behind it, or NULL when the text at *q is anything else (an anchored inline value is
 * y_inline's business) */
static char *y_block_anchor(yparser const *p, size_t *q) {
    if (*q >= p->len || p->s[*q] != '&') return NULL;
    size_t const a = *q + 1;
    size_t e = a;
    while (e < p->len && !isspace((unsigned char)p->s[e])) ++e;
    size_t r = e;
    while (r < p->len && p->s[r] == ' ') ++r;
    if (r < p->len && p->s[r] != '\n' && p->s[r] != '\r' && p->s[r] != '#') return NULL;
    *q = r;
    return y_substr(p->s + a, e - a);
}
static void y_bind_anchor(yparser *p, char *name, ynode *val) {
    if (!name) return;
    if (val && p->n_anchors < 64) { p->anchor_names[p->n_anchors] = name; p->anchor_nodes[p->n_anchors++] = val; return; }
    if (val) y_fail(p, "too many anchors");
    free(name);
}
/* does the text at the cursor read `key: ...` (a mapping entry) rather than a scalar / flow value? */
static int y_looks_like_key(yparser const *p) {
    size_t q = p->pos;
    if (q < p->len && (p->s[q] == '[' || p->s[q] == '{' || p->s[q] == '"' || p->s[q] == '\'' || p->s[q] == '&' || p->s[q] == '*')) return 0;
    while (q < p->len && p->s[q] != '\n' && p->s[q] != '#') {
        if (p->s[q] == ':' && (q + 1 >= p->len || p->s[q + 1] == ' ' || p->s[q + 1] == '\n' || p->s[q + 1] == '\r')) return 1;
        ++q;
    }
    return 0;
}
/* mapping whose entries start at column `ind`; the cursor is at the first key */
static ynode *y_map(yparser *p, int ind) {
    ynode *map = y_new(Y_MAP);
    for (;;) {
        size_t const a = p->pos;
        while (p->pos < p->len && p->s[p->pos] != ':' && p->s[p->pos] != '\n') ++p->pos;
        if (p->pos >= p->len || p->s[p->pos] != ':') { y_free(map); y_fail(p, "expected 'key:'"); return NULL; }
        char *key = y_substr(p->s + a, p->pos - a);
        ++p->pos;
        size_t q = p->pos;
        while (q < p->len && p->s[q] == ' ') ++q;
        ynode *val;
        char *blk_anchor = y_block_anchor(p, &q); /* "key: &name" + a block collection underneath */
        if (q >= p->len || p->s[q] == '\n' || p->s[q] == '\r' || p->s[q] == '#') {
            y_to_line_start(p);
            int const nxt = y_peek_indent(p);
            if (nxt > ind) val = y_block(p, ind + 1);
            else if (nxt == ind) { /* "key:\n- item" at the same indentation is a sequence value */
                size_t const save = p->pos;
                y_next_line(p);
                int const dash = p->pos + 1 < p->len && p->s[p->pos] == '-' && (p->s[p->pos + 1] == ' ' || p->s[p->pos + 1] == '\n');
                p->pos = save;
                val = dash ? y_block(p, ind) : y_new(Y_SCALAR);
            } else val = y_new(Y_SCALAR);
            y_bind_anchor(p, blk_anchor, val);
        } else val = y_inline(p);
        if (!val || p->err[0]) { free(key); y_free(val); y_free(map); return NULL; }
        y_push(map, key, val);
        size_t const save = p->pos;
        int const nxt = y_next_line(p);
        if (nxt != ind || (p->s[p->pos] == '-' && p->pos + 1 < p->len && (p->s[p->pos + 1] == ' ' || p->s[p->pos + 1] == '\n'))) {
            if (nxt > ind && !p->err[0]) { y_free(map); y_fail(p, "unexpected indentation"); return NULL; }
            p->pos = save;
            return map;
        }
    }
}
static ynode *y_seq(yparser *p, int ind) { /* cursor at the '-' of the first item */
    ynode *seq = y_new(Y_SEQ);
    for (;;) {
        ++p->pos; /* '-' */
        int col = ind + 1;
        while (p->pos < p->len && p->s[p->pos] == ' ') { ++p->pos; ++col; }
        ynode *item;
        size_t qa = p->pos;
        char *blk_anchor = y_block_anchor(p, &qa); /* "- &name" + a block collection underneath */
        if (blk_anchor) p->pos = qa;
        if (p->pos >= p->len || p->s[p->pos] == '\n' || p->s[p->pos] == '\r' || p->s[p->pos] == '#') {
            y_to_line_start(p);
            item = y_block(p, ind + 1);
            if (!item && !p->err[0]) item = y_new(Y_SCALAR);
            y_bind_anchor(p, blk_anchor, item);
        } else if (p->s[p->pos] == '-' && p->pos + 1 < p->len && p->s[p->pos + 1] == ' ') item = y_seq(p, col); /* "- - a" */
        else if (y_looks_like_key(p)) item = y_map(p, col);
        else item = y_inline(p);
        if (!item || p->err[0]) { y_free(item); y_free(seq); return NULL; }
        y_push(seq, NULL, item);
        size_t const save = p->pos;
        int const nxt = y_next_line(p);
        if (nxt != ind || !(p->s[p->pos] == '-' && (p->pos + 1 >= p->len || p->s[p->pos + 1] == ' ' || p->s[p->pos + 1] == '\n'))) {
            p->pos = save;
            return seq;
        }
    }
}
static ynode *y_block(yparser *p, int min_indent) {
    size_t const save = p->pos;
    int const ind = y_next_line(p);
    if (ind < 0 || ind < min_indent) { p->pos = save; return NULL; }
    if (p->s[p->pos] == '-' && (p->pos + 1 >= p->len || p->s[p->pos + 1] == ' ' || p->s[p->pos + 1] == '\n')) return y_seq(p, ind);
    if (y_looks_like_key(p)) return y_map(p, ind);
    return y_inline(p);
}

/* ------------------------------------------------------------------------------------------------ */
/* expressions -> non-branching terms                                                                */
/* ------------------------------------------------------------------------------------------------ */
typedef struct { double re, im; } cplx;
static cplx c_mul(cplx a, cplx b) { cplx r = {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; return r; }
/* single-site operator: input bit b -> (output bit, coefficient), or nothing */
typedef struct { int has[2], out[2]; cplx c[2]; } site_op;
static int site_op_of(char kind, double pref, site_op *o) {
    memset(o, 0, sizeof(*o));
    cplx const one = {pref, 0}, mone = {-pref, 0}, pi = {0, pref}, mi = {0, -pref};
    switch (kind) {
    case 'x': o->has[0] = o->has[1] = 1; o->out[0] = 1; o->out[1] = 0; o->c[0] = one; o->c[1] = one; return 0;
    case 'y': o->has[0] = o->has[1] = 1; o->out[0] = 1; o->out[1] = 0; o->c[0] = pi; o->c[1] = mi; return 0;
    case 'z': o->has[0] = o->has[1] = 1; o->out[0] = 0; o->out[1] = 1; o->c[0] = one; o->c[1] = mone; return 0;
    case '+': o->has[1] = 1; o->out[1] = 0; o->c[1] = one; return 0; /* |up><down| */
    case '-': o->has[0] = 1; o->out[0] = 1; o->c[0] = one; return 0;
    case 'I': o->has[0] = o->has[1] = 1; o->out[0] = 0; o->out[1] = 1; o->c[0] = one; o->c[1] = one; return 0;
    default: return -1;
    }
}
/* `second` applied after `first` */
static site_op site_compose(site_op const *first, site_op const *second) {
    site_op r;
    memset(&r, 0, sizeof(r));
    for (int b = 0; b < 2; ++b)
        if (first->has[b] && second->has[first->out[b]]) {
            r.has[b] = 1;
            r.out[b] = second->out[first->out[b]];
            r.c[b] = c_mul(first->c[b], second->c[first->out[b]]);
        }
    return r;
}
/* mutually exclusive alternatives of a site operator: Pauli-like ones need no projector */
typedef struct { int need, rbit, flip, sign; cplx c; } site_alt;
static int site_alternatives(site_op const *o, site_alt alts[2]) {
    if (o->has[0] && o->has[1]) {
        int const flip0 = o->out[0] != 0, flip1 = o->out[1] != 1;
        if (flip0 == flip1 && o->c[1].re == o->c[0].re && o->c[1].im == o->c[0].im) {
            site_alt a = {0, 0, flip0, 0, o->c[0]};
            alts[0] = a;
            return 1;
        }
        if (flip0 == flip1 && o->c[1].re == -o->c[0].re && o->c[1].im == -o->c[0].im) {
            site_alt a = {0, 0, flip0, 1, o->c[0]};
            alts[0] = a;
            return 1;
        }
    }
    int n = 0;
    for (int b = 0; b < 2; ++b)
        if (o->has[b]) {
            site_alt a = {1, b, o->out[b] != b, 0, o->c[b]};
            alts[n++] = a;
        }
    return n;
}
/* UTF-8 helpers for "σˣ₀" */
static int utf8_starts(char const *s, char const *lit) { return strncmp(s, lit, strlen(lit)) == 0; }

#define MAX_FACTORS 16
typedef struct { char kind; int idx; double pref; } factor;
/* '0.8 × σˣ₀ σˣ₁' -> scalar, factors */
static int parse_expression(char const *expr, cplx *scalar, factor *f, int *nf) {
    scalar->re = 1.0; scalar->im = 0.0;
    *nf = 0;
    char const *p = expr;
    while (*p) {
        while (*p == ' ' || *p == '\t' || *p == '*') ++p;
        if (utf8_starts(p, "\xc3\x97")) { p += 2; continue; } /* the multiplication sign */
        if (!*p) break;
        double pref = 0.0;
        if (utf8_starts(p, "\xcf\x83")) { pref = 1.0; p += 2; }     /* sigma */
        else if (*p == 'S') { pref = 0.5; p += 1; }
        if (pref != 0.0) {
            char kind = 0;
            if (utf8_starts(p, "\xcb\xa3")) { kind = 'x'; p += 2; }
            else if (utf8_starts(p, "\xca\xb8")) { kind = 'y'; p += 2; }
            else if (utf8_starts(p, "\xe1\xb6\xbb")) { kind = 'z'; p += 3; }
            else if (utf8_starts(p, "\xe2\x81\xba")) { kind = '+'; p += 3; }
            else if (utf8_starts(p, "\xe2\x81\xbb")) { kind = '-'; p += 3; }
            else return ls_amd_internal_error("cannot parse an operator in the expression '%s'", expr);
            int idx = 0, digits = 0;
            while ((unsigned char)p[0] == 0xe2 && (unsigned char)p[1] == 0x82 && (unsigned char)p[2] >= 0x80 && (unsigned char)p[2] <= 0x89) {
                idx = idx * 10 + ((unsigned char)p[2] - 0x80);
                p += 3;
                if (++digits > 6) return ls_amd_internal_error("site index too long in the expression '%s'", expr);
            }
            if (!digits || (*p && *p != ' ' && *p != '\t')) return ls_amd_internal_error("cannot parse a site index in the expression '%s'", expr);
            if (*nf >= MAX_FACTORS) return ls_amd_internal_error("too many factors in the expression '%s'", expr);
            f[*nf].kind = kind; f[*nf].idx = idx; f[*nf].pref = pref;
            ++*nf;
        } else {
            char *end;
            double const v = strtod(p, &end);
            if (end == p) return ls_amd_internal_error("cannot parse '%s' in the expression '%s'", p, expr);
            cplx s = {v, 0.0};
            if (*end == 'j' || *end == 'i') { s.re = 0.0; s.im = v; ++end; }
            *scalar = c_mul(*scalar, s);
            p = end;
        }
    }
    if (*nf == 0) return ls_amd_internal_error("the expression '%s' has no operators", expr);
    return 0;
}

typedef struct { int n, cap; double *v; uint64_t *m, *r, *x, *s; } term_list;
static void terms_push(term_list *t, cplx v, uint64_t m, uint64_t r, uint64_t x, uint64_t s) {
    if (t->n == t->cap) {
        t->cap = t->cap ? 2 * t->cap : 64;
        t->v = (double *)realloc(t->v, sizeof(double) * 2 * (size_t)t->cap);
        t->m = (uint64_t *)realloc(t->m, sizeof(uint64_t) * (size_t)t->cap);
        t->r = (uint64_t *)realloc(t->r, sizeof(uint64_t) * (size_t)t->cap);
        t->x = (uint64_t *)realloc(t->x, sizeof(uint64_t) * (size_t)t->cap);
        t->s = (uint64_t *)realloc(t->s, sizeof(uint64_t) * (size_t)t->cap);
    }
    t->v[2 * t->n] = v.re; t->v[2 * t->n + 1] = v.im;
    t->m[t->n] = m; t->r[t->n] = r; t->x[t->n] = x; t->s[t->n] = s;
    ++t->n;
}
static void terms_free(term_list *t) { free(t->v); free(t->m); free(t->r); free(t->x); free(t->s); }

/* terms of one monomial on one tuple of global sites */
static int monomial_terms(char const *expr, int const *sites, int n_sites, int number_sites, term_list *out) {
    cplx scalar;
    factor f[MAX_FACTORS];
    int nf;
    if (parse_expression(expr, &scalar, f, &nf) != 0) return -1;
    /* operators written left to right act right to left on a ket */
    site_op per[MAX_FACTORS];
    int order[MAX_FACTORS], n_order = 0, max_idx = -1;
    for (int q = nf - 1; q >= 0; --q) {
        site_op o;
        site_op_of(f[q].kind, f[q].pref, &o);
        int pos = -1;
        for (int k = 0; k < n_order; ++k) if (order[k] == f[q].idx) pos = k;
        if (pos >= 0) per[pos] = site_compose(&per[pos], &o);
        else { per[n_order] = o; order[n_order++] = f[q].idx; }
        if (f[q].idx > max_idx) max_idx = f[q].idx;
    }
    if (n_sites != max_idx + 1) return ls_amd_internal_error("the expression '%s' needs %d sites, got %d", expr, max_idx + 1, n_sites);
    for (int i = 0; i < n_sites; ++i) {
        if (sites[i] < 0 || sites[i] >= number_sites) return ls_amd_internal_error("site %d out of range in the sites of '%s'", sites[i], expr);
        for (int j = 0; j < i; ++j) if (sites[i] == sites[j]) return ls_amd_internal_error("repeated site %d in the sites of '%s'", sites[i], expr);
    }
    site_alt alts[MAX_FACTORS][2];
    int n_alts[MAX_FACTORS], choice[MAX_FACTORS];
    for (int k = 0; k < n_order; ++k) { n_alts[k] = site_alternatives(&per[k], alts[k]); choice[k] = 0; if (n_alts[k] == 0) return 0; }
    for (;;) { /* the product of the alternatives */
        cplx v = scalar;
        uint64_t m = 0, r = 0, x = 0, s = 0;
        for (int k = 0; k < n_order; ++k) {
            site_alt const *a = &alts[k][choice[k]];
            uint64_t const bit = 1ULL << sites[order[k]];
            v = c_mul(v, a->c);
            if (a->need) { m |= bit; if (a->rbit) r |= bit; }
            if (a->flip) x |= bit;
            if (a->sign) s |= bit;
        }
        if (v.re != 0.0 || v.im != 0.0) terms_push(out, v, m, r, x, s);
        int k = n_order - 1;
        while (k >= 0 && ++choice[k] == n_alts[k]) choice[k--] = 0;
        if (k < 0) break;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------ */
/* document -> objects                                                                              */
/* ------------------------------------------------------------------------------------------------ */
static int y_int(ynode const *n, char const *what, long *out) {
    if (!n || n->kind != Y_SCALAR || !n->str || n->quoted) return ls_amd_internal_error("%s: expected an integer", what);
    char *end;
    long const v = strtol(n->str, &end, 10);
    if (end == n->str || *end) return ls_amd_internal_error("%s: '%s' is not an integer", what, n->str);
    /* every integer of a config ends up in an `int` (sites, permutation entries, sectors, weights): 4294967296 must not read as 0 */
    if (v < -2147483647L - 1 || v > 2147483647L) return ls_amd_internal_error("%s: %s is out of range", what, n->str);
    *out = v;
    return 0;
}
static ls_hs_basis *basis_from(ynode const *b) {
    if (!b || b->kind != Y_MAP) { ls_amd_internal_error("the config has no `basis` section"); return NULL; }
    long L, hw = -1, inv = 0;
    if (y_int(y_get(b, "number_spins"), "basis.number_spins", &L) != 0) return NULL;
    if (L < 1 || L > 64) { ls_amd_internal_error("basis.number_spins = %ld: 1 .. 64 sites are supported", L); return NULL; }
    if (!y_is_null(y_get(b, "hamming_weight")) && y_int(y_get(b, "hamming_weight"), "basis.hamming_weight", &hw) != 0) return NULL;
    if (!y_is_null(y_get(b, "spin_inversion")) && y_int(y_get(b, "spin_inversion"), "basis.spin_inversion", &inv) != 0) return NULL;
    ynode const *particle = y_get(b, "particle");
    if (particle && !y_is_null(particle) && (particle->kind != Y_SCALAR || !particle->str || strcmp(particle->str, "spin-1/2") != 0)) {
        ls_amd_internal_error("only spin-1/2 bases are supported, got '%s'", particle->kind == Y_SCALAR && particle->str ? particle->str : "<a collection>");
        return NULL;
    }
    ynode const *syms = y_get(b, "symmetries");
    int ng = 0;
    int *perms = NULL, *sectors = NULL;
    if (syms && !y_is_null(syms)) {
        if (syms->kind != Y_SEQ) { ls_amd_internal_error("basis.symmetries: expected a sequence"); return NULL; }
        ng = syms->n;
        perms = (int *)calloc((size_t)(ng > 0 ? ng : 1) * (size_t)L, sizeof(int));
        sectors = (int *)calloc((size_t)(ng > 0 ? ng : 1), sizeof(int));
        for (int g = 0; g < ng; ++g) {
            ynode const *perm = y_get(syms->vals[g], "permutation");
            long sec = 0;
            int bad = !perm || perm->kind != Y_SEQ || perm->n != L || y_int(y_get(syms->vals[g], "sector"), "symmetries[].sector", &sec) != 0;
            for (int i = 0; i < L && !bad; ++i) {
                long v = 0;
                bad = y_int(perm->vals[i], "symmetries[].permutation", &v) != 0;
                perms[(size_t)g * (size_t)L + i] = (int)v;
            }
            if (bad) {
                free(perms); free(sectors);
                ls_amd_internal_error("basis.symmetries[%d]: expected {permutation: [%ld sites], sector: int}", g, L);
                return NULL;
            }
            sectors[g] = (int)sec;
        }
    }
    ls_hs_basis *basis = ls_hs_create_spin_basis((int)L, (int)hw, (int)inv, ng, perms, sectors);
    free(perms); free(sectors);
    return basis;
}
static ls_hs_operator *operator_from(ls_hs_basis const *basis, ynode const *section, char const *what) {
    ynode const *terms = y_get(section, "terms");
    if (!terms || terms->kind != Y_SEQ) { ls_amd_internal_error("%s: expected a `terms` sequence", what); return NULL; }
    term_list tl;
    memset(&tl, 0, sizeof(tl));
    for (int t = 0; t < terms->n; ++t) {
        ynode const *expr = y_get(terms->vals[t], "expression"), *sites = y_get(terms->vals[t], "sites");
        if (!expr || expr->kind != Y_SCALAR || !expr->str) {
            terms_free(&tl);
            ls_amd_internal_error("%s.terms[%d]: only the `expression:` schema is supported (old-schema `matrix:` files are inputs of "
                                  "input_for_matvec.py only)", what, t);
            return NULL;
        }
        if (!sites || sites->kind != Y_SEQ) { terms_free(&tl); ls_amd_internal_error("%s.terms[%d]: expected `sites: [[...], ...]`", what, t); return NULL; }
        for (int q = 0; q < sites->n; ++q) {
            ynode const *tuple = sites->vals[q];
            int idx[MAX_FACTORS];
            int bad = !tuple || tuple->kind != Y_SEQ || tuple->n > MAX_FACTORS;
            for (int i = 0; !bad && i < tuple->n; ++i) {
                long v = 0;
                bad = y_int(tuple->vals[i], "sites", &v) != 0;
                idx[i] = (int)v;
            }
            if (bad || monomial_terms(expr->str, idx, tuple->n, basis->number_sites, &tl) != 0) {
                if (bad) ls_amd_internal_error("%s.terms[%d].sites[%d]: expected a tuple of site indices", what, t, q);
                terms_free(&tl);
                return NULL;
            }
        }
    }
    ls_hs_operator *op = ls_hs_create_operator_from_terms(basis, tl.n, tl.v, tl.m, tl.r, tl.x, tl.s);
    terms_free(&tl);
    return op;
}

void ls_hs_destroy_yaml_config(ls_hs_yaml_config *conf) {
    if (!conf) return;
    for (int i = 0; i < conf->number_observables; ++i) ls_hs_destroy_operator(conf->observables[i]);
    free(conf->observables);
    if (conf->hamiltonian) ls_hs_destroy_operator(conf->hamiltonian);
    if (conf->basis) ls_hs_destroy_basis(conf->basis);
    free(conf);
}

/* the same from a NUL-terminated YAML text (what the file loader reads) */
ls_hs_yaml_config *ls_amd_load_yaml_config_from_string(char const *text) {
    yparser p;
    memset(&p, 0, sizeof(p));
    p.s = text;
    p.len = strlen(text);
    ynode *doc = y_block(&p, 0);
    for (int i = 0; i < p.n_anchors; ++i) free(p.anchor_names[i]);
    if (!doc || p.err[0] || doc->kind != Y_MAP) {
        ls_amd_internal_error("YAML: %s", p.err[0] ? p.err : "the document is not a mapping");
        y_free(doc);
        return NULL;
    }
    ls_hs_yaml_config *conf = (ls_hs_yaml_config *)calloc(1, sizeof(*conf));
    int ok = (conf->basis = basis_from(y_get(doc, "basis"))) != NULL;
    ynode const *h = y_get(doc, "hamiltonian");
    if (ok && h && !y_is_null(h)) ok = (conf->hamiltonian = operator_from(conf->basis, h, "hamiltonian")) != NULL;
    ynode const *obs = y_get(doc, "observables");
    if (ok && obs && !y_is_null(obs)) {
        if (obs->kind != Y_SEQ) { ls_amd_internal_error("observables: expected a sequence"); ok = 0; }
        else {
            conf->observables = (ls_hs_operator **)calloc((size_t)(obs->n > 0 ? obs->n : 1), sizeof(ls_hs_operator *));
            for (int i = 0; ok && i < obs->n; ++i) {
                ok = (conf->observables[i] = operator_from(conf->basis, obs->vals[i], "observables[]")) != NULL;
                if (ok) conf->number_observables = i + 1;
            }
        }
    }
    y_free(doc);
    if (!ok) { ls_hs_destroy_yaml_config(conf); return NULL; }
    ls_amd_internal_clear_error(); /* success: no stale message of an earlier failed load */
    return conf;
}

/* /root/reference/src/FFI.chpl:208: NULL on failure (ls_amd_last_error() says why; the Chapel caller halts with
 * "failed to load Config from '<file>'", ForeignTypes.chpl:264-265) */
ls_hs_yaml_config *ls_hs_load_yaml_config(char const *filename) {
    FILE *f = fopen(filename, "rb");
    if (!f) { ls_amd_internal_error("cannot open '%s'", filename); return NULL; }
    fseek(f, 0, SEEK_END);
    long const n = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *text = (char *)malloc((size_t)(n > 0 ? n : 0) + 1);
    size_t const got = n > 0 ? fread(text, 1, (size_t)n, f) : 0;
    fclose(f);
    text[got] = 0;
    ls_hs_yaml_config *conf = ls_amd_load_yaml_config_from_string(text);
    free(text);
    return conf;
}
