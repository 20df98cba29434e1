// clipfile_host.cpp - the reference's on-disk clip files read by native host code, straight into the model's input tensors.
//
//   <video>.pkl      written by reference baselines/preprocess_perception_main.py:87-96 (pickle.HIGHEST_PROTOCOL) and read back by
//                    baselines/datasets.py:60-64, 583-587:  {"bb": list[T] of int ndarray [n_t, 4] (x1, y1, x2, y2 pixels),
//                    "labels": list[T] of int ndarray [n_t] (class ids)} (+ any further keys, e.g. "3d_coord" of the perfect-
//                    perception generator, which the datasets never look at)
//   <video>_bb.json  written by generate/render_videos.py:436-457, read by baselines/datasets.py:33-45:
//                    {"<object name>": list[T] of [x, y, w, h], ...}; only "small_gold_spl_metal_Spl_0" (the snitch) is used:
//                    labels[t] = [x, y, x + w, y + h] / [320, 240, 320, 240] in float64, then torch.tensor(dtype=float32)
//
// Why: pickle.load of the 600 small arrays of one clip is 0.4 ms and json.load 0.07 ms of a 0.55 ms sample, against 16 us for the
// encode itself - the input side, not the GPU, bounds the inference driver.
//
// The pickle reader is a RESTRICTED unpickler: a small stack machine over exactly the opcodes protocols 2-5 emit for "dict of
// lists of C-contiguous numeric ndarrays", with a whitelist of three callables (numpy's _reconstruct, dtype and _frombuffer under
// numpy.core / numpy._core).  Nothing is imported, nothing is executed; anything else - another global, an object array, a
// Fortran-ordered or big-endian array, an out-of-band buffer, a truncated stream - is REFUSED with a message naming the file and
// the offending opcode, never guessed at.  No heap object of the stream outlives the call.
#ifdef __clang__
#pragma clang fp contract(off)
#endif

#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <deque>
#include <string>
#include <thread>
#include <vector>

extern "C" int opnet_encode_clips_f32(const int32_t *counts, const int32_t *ids, const int32_t *bb, const int64_t *clip_first_frame,
                                      const int64_t *clip_first_det, int n_clips, int T, int n_tracks, const uint8_t *is_cone,
                                      int n_classes, float *boxes_out, int64_t *index_out);

namespace {

struct Err {
    char *buf;
    int len;
    int fail(int code, const char *fmt, ...) const __attribute__((format(printf, 3, 4)))
    {
        if (buf && len > 0) {
            va_list ap;
            va_start(ap, fmt);
            vsnprintf(buf, (size_t)len, fmt, ap);
            va_end(ap);
        }
        return code;
    }
};

bool read_file(const char *path, std::vector<unsigned char> &out)
{
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    out.clear();
    if (fseek(f, 0, SEEK_END) == 0) {
        const long n = ftell(f);
        if (n > 0) out.resize((size_t)n);
        fseek(f, 0, SEEK_SET);
    }
    size_t got = 0;
    if (!out.empty()) got = fread(out.data(), 1, out.size(), f);
    else {                                   // not seekable: read in pieces
        unsigned char tmp[65536];
        size_t n;
        while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) out.insert(out.end(), tmp, tmp + n);
        got = out.size();
    }
    fclose(f);
    out.resize(got);
    return true;
}

// ---- the restricted unpickler ------------------------------------------------------------------------------------------------
enum Kind : uint8_t { K_NONE, K_BOOL, K_INT, K_FLOAT, K_STR, K_BYTES, K_TUPLE, K_LIST, K_DICT, K_GLOBAL, K_DTYPE, K_RECON, K_ARRAY, K_MARK };
enum Glob : uint8_t { G_OTHER, G_RECONSTRUCT, G_NDARRAY, G_DTYPE, G_FROMBUFFER, G_LATIN1, G_BYTES };

struct Val {                         // plain data, 56 bytes
    Kind k = K_NONE;
    Glob g = G_OTHER;
    char tkind = 0;                  // K_DTYPE / K_ARRAY: numpy kind character 'i' 'u' 'f' 'b'
    char border = '|';               // '<' '|' '=' ok; '>' refused for multi-byte items
    uint8_t tsize = 0;               // item size in bytes
    uint8_t ndim = 0;
    int kid0 = 0, nkid = 0;          // K_TUPLE / K_LIST: items; K_DICT: key, value, key, value ... = Unpickler::kids[kid0 .. kid0 + nkid)
    long long i = 0;                 // K_INT / K_BOOL
    const unsigned char *p = nullptr;   // K_STR / K_BYTES: into the file image; K_ARRAY: the data
    size_t n = 0;                    // byte length
    int shape[4] = {0, 0, 0, 0};
};

struct Unpickler {
    const unsigned char *d;
    size_t n, pos = 0;
    const char *path;
    Err err;
    // (references to buffers the caller keeps across files: a fresh 1.5 MB pool per clip is 370 page faults, more than the parse)
    std::vector<Val> &pool;
    std::vector<int> &stack, &memo;
    std::vector<int> &kids;                // the children of every container, one range per container (Val is plain data: the
                                           // 600 arrays of a clip are ~8 000 values - a vector in each cost more than the parse)
    std::deque<std::string> decoded;       // (a deque: push_back never moves an element) byte strings rebuilt from protocol 2's latin-1 text (owned here, Val.p points in)

    int push(const Val &v) { pool.push_back(v); stack.push_back((int)pool.size() - 1); return 0; }
    int kid(const Val &v, int q) const { return kids[v.kid0 + q]; }
    // container v takes the stack entries [from, end) as (further) children
    void adopt(Val &v, size_t from)
    {
        const int add = (int)(stack.size() - from);
        if (v.nkid > 0 && v.kid0 + v.nkid != (int)kids.size()) {       // not the last range: move it to the end first
            const int k0 = (int)kids.size();
            kids.resize(kids.size() + v.nkid);
            for (int q = 0; q < v.nkid; ++q) kids[k0 + q] = kids[v.kid0 + q];
            v.kid0 = k0;
        } else if (v.nkid == 0) {
            v.kid0 = (int)kids.size();
        }
        kids.insert(kids.end(), stack.begin() + from, stack.end());
        v.nkid += add;
        stack.resize(from);
    }
    bool need(size_t k) const { return pos + k <= n; }
    unsigned long long le(size_t k)
    {
        unsigned long long v = 0;
        for (size_t b = 0; b < k; ++b) v |= (unsigned long long)d[pos + b] << (8 * b);
        pos += k;
        return v;
    }
    int bad(const char *what, unsigned op = 0) { return err.fail(-6, "%s: %s (opcode 0x%02x at byte %zu)", path, what, op, pos); }

    static bool streq(const Val &v, const char *s) { return v.k == K_STR && v.n == strlen(s) && memcmp(v.p, s, v.n) == 0; }

    Glob classify(const Val &mod, const Val &name)
    {
        const bool core_multi = streq(mod, "numpy.core.multiarray") || streq(mod, "numpy._core.multiarray");
        const bool core_num = streq(mod, "numpy.core.numeric") || streq(mod, "numpy._core.numeric");
        if (core_multi && streq(name, "_reconstruct")) return G_RECONSTRUCT;
        if (streq(mod, "numpy") && streq(name, "ndarray")) return G_NDARRAY;
        if (streq(mod, "numpy") && streq(name, "dtype")) return G_DTYPE;
        if (core_num && streq(name, "_frombuffer")) return G_FROMBUFFER;
        // protocol 2 has no bytes opcode: array data travels as _codecs.encode(<latin-1 text>, "latin1"), b"" as bytes()
        if (streq(mod, "_codecs") && streq(name, "encode")) return G_LATIN1;
        if ((streq(mod, "__builtin__") || streq(mod, "builtins")) && streq(name, "bytes")) return G_BYTES;
        return G_OTHER;
    }

    // index of the first stack entry above the topmost MARK (0 = there is none)
    size_t above_mark() const
    {
        size_t m = stack.size();
        while (m > 0 && pool[stack[m - 1]].k != K_MARK) --m;
        return m;
    }

    int parse_dtype_desc(const Val &s, Val &out)
    {
        // numpy.dtype('<kind><size>', False, True): 'i8', 'i4', 'u1', 'f4', 'b1' ...
        if (s.k != K_STR || s.n < 2 || s.n > 3) return bad("unsupported dtype description");
        out.tkind = (char)s.p[0];
        out.tsize = (uint8_t)atoi(std::string((const char *)s.p + 1, s.n - 1).c_str());
        if (!strchr("iufb", out.tkind) || (out.tsize != 1 && out.tsize != 2 && out.tsize != 4 && out.tsize != 8))
            return bad("only fixed-size numeric arrays are read (object / string / structured dtypes are refused)");
        return 0;
    }

    int set_shape(Val &a, const Val &shape)
    {
        if (shape.k != K_TUPLE || shape.nkid > 4) return bad("array shape must be a tuple of at most 4 ints");
        a.ndim = (uint8_t)shape.nkid;
        long long total = 1;
        for (int q = 0; q < a.ndim; ++q) {
            const Val &e = pool[kid(shape, q)];
            if (e.k != K_INT || e.i < 0 || e.i > 0x7fffffff) return bad("array shape must be a tuple of non-negative ints");
            a.shape[q] = (int)e.i;
            total *= e.i;
        }
        if ((unsigned long long)total * (unsigned long long)a.tsize != a.n) return bad("array data length does not match its shape");
        return 0;
    }

    int run()
    {
        while (true) {
            if (!need(1)) return bad("truncated stream");
            const unsigned op = d[pos++];
            switch (op) {
            case 0x80: if (!need(1)) return bad("truncated", op); if (d[pos] < 2 || d[pos] > 5) return bad("pickle protocol must be 2..5", op); ++pos; break;
            case 0x95: if (!need(8)) return bad("truncated", op); pos += 8; break;                                   // FRAME
            case '}': { Val v; v.k = K_DICT; push(v); break; }
            case ']': { Val v; v.k = K_LIST; push(v); break; }
            case ')': { Val v; v.k = K_TUPLE; push(v); break; }
            case '(': { Val v; v.k = K_MARK; push(v); break; }
            case 'N': { Val v; push(v); break; }
            case 0x88: case 0x89: { Val v; v.k = K_BOOL; v.i = op == 0x88; push(v); break; }
            case 'K': case 'M': case 'J': {
                const size_t w = op == 'K' ? 1 : op == 'M' ? 2 : 4;
                if (!need(w)) return bad("truncated", op);
                Val v; v.k = K_INT;
                const unsigned long long u = le(w);
                v.i = op == 'J' ? (long long)(int32_t)(uint32_t)u : (long long)u;
                push(v);
                break;
            }
            case 0x8a: {                                                                                         // LONG1
                if (!need(1)) return bad("truncated", op);
                const size_t w = d[pos++];
                if (w > 8 || !need(w)) return bad("integer wider than 64 bits", op);
                Val v; v.k = K_INT;
                unsigned long long u = le(w);
                if (w > 0 && w < 8 && (u >> (8 * w - 1)) & 1) u |= ~0ULL << (8 * w);
                v.i = (long long)u;
                push(v);
                break;
            }
            case 'G': { if (!need(8)) return bad("truncated", op); Val v; v.k = K_FLOAT; pos += 8; push(v); break; }
            case 0x8c: case 'X': case 0x8d: case 'C': case 'B': case 0x8e: case 0x96: {
                const size_t w = (op == 0x8c || op == 'C') ? 1 : (op == 'X' || op == 'B') ? 4 : 8;
                if (!need(w)) return bad("truncated", op);
                const unsigned long long len = le(w);
                if (len > n || !need((size_t)len)) return bad("string / bytes run past the end of the file", op);
                Val v; v.k = (op == 0x8c || op == 'X' || op == 0x8d) ? K_STR : K_BYTES;
                v.p = d + pos; v.n = (size_t)len;
                pos += (size_t)len;
                push(v);
                break;
            }
            case 0x98: break;                                                                                    // READONLY_BUFFER
            case 0x97: return bad("out-of-band buffers are not supported", op);                                   // NEXT_BUFFER
            case 0x94: if (stack.empty()) return bad("MEMOIZE on an empty stack", op); memo.push_back(stack.back()); break;
            case 'q': case 'r': {
                const size_t w = op == 'q' ? 1 : 4;
                if (!need(w) || stack.empty()) return bad("truncated", op);
                const size_t idx = (size_t)le(w);
                if (idx > (1u << 24)) return bad("memo index out of range", op);
                if (memo.size() <= idx) memo.resize(idx + 1, -1);
                memo[idx] = stack.back();
                break;
            }
            case 'h': case 'j': {
                const size_t w = op == 'h' ? 1 : 4;
                if (!need(w)) return bad("truncated", op);
                const size_t idx = (size_t)le(w);
                if (idx >= memo.size() || memo[idx] < 0) return bad("memo entry does not exist", op);
                stack.push_back(memo[idx]);
                break;
            }
            case 0x85: case 0x86: case 0x87: {
                const size_t c = op - 0x84;
                if (stack.size() < c) return bad("stack underflow", op);
                Val v; v.k = K_TUPLE;
                adopt(v, stack.size() - c);
                push(v);
                break;
            }
            case 't': case 'l': {
                const size_t m = above_mark();
                if (m == 0) return bad("no MARK on the stack", op);
                Val v; v.k = op == 't' ? K_TUPLE : K_LIST;
                adopt(v, m);
                stack.pop_back();                                   // the MARK
                push(v);
                break;
            }
            case 'a': case 's': {                                   // APPEND (1 item) / SETITEM (key, value)
                const size_t c = op == 'a' ? 1 : 2;
                if (stack.size() < c + 1) return bad("stack underflow", op);
                const int owner = stack[stack.size() - c - 1];
                if (pool[owner].k != (op == 'a' ? K_LIST : K_DICT)) return bad("APPEND / SETITEM on the wrong kind of object", op);
                adopt(pool[owner], stack.size() - c);
                break;
            }
            case 'e': case 'u': {                                   // APPENDS / SETITEMS: everything above the MARK
                const size_t m = above_mark();
                if (m < 2) return bad("no MARK on the stack", op);
                const int owner = stack[m - 2];
                if (pool[owner].k != (op == 'e' ? K_LIST : K_DICT) || (op == 'u' && ((stack.size() - m) & 1)))
                    return bad("APPENDS / SETITEMS on the wrong kind of object", op);
                adopt(pool[owner], m);
                stack.pop_back();                                   // the MARK
                break;
            }
            case 'c': {                                                                                          // GLOBAL: "module\nname\n"
                Val mod, name;
                mod.k = name.k = K_STR;
                const unsigned char *e = (const unsigned char *)memchr(d + pos, '\n', n - pos);
                if (!e) return bad("truncated", op);
                mod.p = d + pos; mod.n = (size_t)(e - (d + pos)); pos += mod.n + 1;
                e = (const unsigned char *)memchr(d + pos, '\n', n - pos);
                if (!e) return bad("truncated", op);
                name.p = d + pos; name.n = (size_t)(e - (d + pos)); pos += name.n + 1;
                Val v; v.k = K_GLOBAL; v.g = classify(mod, name);
                if (v.g == G_OTHER) return err.fail(-6, "%s: refusing to resolve global %.*s.%.*s (only numpy's ndarray / dtype reconstruction is read)",
                                                    path, (int)mod.n, mod.p, (int)name.n, name.p);
                push(v);
                break;
            }
            case 0x93: {                                                                                         // STACK_GLOBAL
                if (stack.size() < 2) return bad("stack underflow", op);
                const Val name = pool[stack.back()]; stack.pop_back();
                const Val mod = pool[stack.back()]; stack.pop_back();
                Val v; v.k = K_GLOBAL; v.g = classify(mod, name);
                if (v.g == G_OTHER)
                    return err.fail(-6, "%s: refusing to resolve global %.*s.%.*s (only numpy's ndarray / dtype reconstruction is read)", path,
                                    mod.k == K_STR ? (int)mod.n : 1, mod.k == K_STR ? (const char *)mod.p : "?",
                                    name.k == K_STR ? (int)name.n : 1, name.k == K_STR ? (const char *)name.p : "?");
                push(v);
                break;
            }
            case 'R': {
                if (stack.size() < 2) return bad("stack underflow", op);
                const Val args = pool[stack.back()]; stack.pop_back();
                const Val fn = pool[stack.back()]; stack.pop_back();
                if (fn.k != K_GLOBAL || args.k != K_TUPLE) return bad("REDUCE of something that is not a whitelisted callable", op);
                Val v;
                if (fn.g == G_RECONSTRUCT) {
                    if (args.nkid != 3 || pool[kid(args, 0)].k != K_GLOBAL || pool[kid(args, 0)].g != G_NDARRAY)
                        return bad("_reconstruct of something that is not numpy.ndarray", op);
                    v.k = K_RECON;
                } else if (fn.g == G_DTYPE) {
                    if (args.nkid == 0) return bad("numpy.dtype() without a description", op);
                    v.k = K_DTYPE;
                    if (int rc = parse_dtype_desc(pool[kid(args, 0)], v)) return rc;
                } else if (fn.g == G_FROMBUFFER) {
                    // _frombuffer(buf, dtype, shape, order)
                    if (args.nkid != 4) return bad("_frombuffer with an unexpected argument list", op);
                    const Val &buf = pool[kid(args, 0)], &dt = pool[kid(args, 1)], &ord = pool[kid(args, 3)];
                    if (buf.k != K_BYTES || dt.k != K_DTYPE) return bad("_frombuffer with an unexpected argument list", op);
                    v.k = K_ARRAY; v.p = buf.p; v.n = buf.n; v.tkind = dt.tkind; v.tsize = dt.tsize; v.border = dt.border;
                    if (int rc = set_shape(v, pool[kid(args, 2)])) return rc;
                    if (!(streq(ord, "C") || v.ndim <= 1)) return bad("only C-contiguous arrays are read", op);
                } else if (fn.g == G_LATIN1) {
                    // _codecs.encode(text, "latin1"): every code point < 256 is one byte
                    if (args.nkid != 2 || pool[kid(args, 0)].k != K_STR || !streq(pool[kid(args, 1)], "latin1"))
                        return bad("_codecs.encode is only read as (text, \"latin1\")", op);
                    const Val &t = pool[kid(args, 0)];
                    std::string out;
                    out.reserve(t.n);
                    for (size_t q = 0; q < t.n; ++q) {
                        const unsigned c = t.p[q];
                        if (c < 0x80) out.push_back((char)c);
                        else if ((c == 0xc2 || c == 0xc3) && q + 1 < t.n && (t.p[q + 1] & 0xc0) == 0x80)
                            out.push_back((char)(((c & 3) << 6) | (t.p[++q] & 0x3f)));
                        else return bad("text that is not latin-1", op);
                    }
                    decoded.push_back(std::move(out));
                    v.k = K_BYTES; v.p = (const unsigned char *)decoded.back().data(); v.n = decoded.back().size();
                } else if (fn.g == G_BYTES) {
                    if (args.nkid != 0) return bad("bytes() is only read without arguments", op);
                    v.k = K_BYTES; v.p = d; v.n = 0;
                } else {
                    return bad("REDUCE of something that is not a whitelisted callable", op);
                }
                if (v.k == K_ARRAY && v.border == '>' && v.tsize > 1) return bad("big-endian arrays are not read", op);
                push(v);
                break;
            }
            case 'b': {
                if (stack.size() < 2) return bad("stack underflow", op);
                const Val st = pool[stack.back()]; stack.pop_back();
                Val &obj = pool[stack.back()];
                if (st.k != K_TUPLE) return bad("BUILD with a state that is not a tuple", op);
                if (obj.k == K_DTYPE) {
                    // (version, byteorder, subdescr, names, fields, itemsize, alignment, flags)
                    if (st.nkid < 2 || pool[kid(st, 1)].k != K_STR || pool[kid(st, 1)].n != 1) return bad("unexpected dtype state", op);
                    obj.border = (char)pool[kid(st, 1)].p[0];
                    if (st.nkid >= 5 && (pool[kid(st, 2)].k != K_NONE || pool[kid(st, 3)].k != K_NONE || pool[kid(st, 4)].k != K_NONE))
                        return bad("structured dtypes are not read", op);
                    if (obj.border == '>' && obj.tsize > 1) return bad("big-endian arrays are not read", op);
                } else if (obj.k == K_RECON) {
                    // (version, shape, dtype, is_fortran, rawdata)
                    if (st.nkid != 5) return bad("unexpected ndarray state", op);
                    const Val &dt = pool[kid(st, 2)], &fo = pool[kid(st, 3)], &raw = pool[kid(st, 4)];
                    if (dt.k != K_DTYPE || raw.k != K_BYTES) return bad("unexpected ndarray state (object arrays are refused)", op);
                    Val a;
                    a.k = K_ARRAY; a.p = raw.p; a.n = raw.n; a.tkind = dt.tkind; a.tsize = dt.tsize; a.border = dt.border;
                    if (int rc = set_shape(a, pool[kid(st, 1)])) return rc;
                    if (fo.k == K_BOOL && fo.i && a.ndim > 1) return bad("only C-contiguous arrays are read", op);
                    if (a.border == '>' && a.tsize > 1) return bad("big-endian arrays are not read", op);
                    obj = a;
                } else {
                    return bad("BUILD on something that is neither a dtype nor an ndarray", op);
                }
                break;
            }
            case '.': return 0;
            default: return bad("opcode outside the subset a clip file uses", op);
            }
        }
    }
};

long long element(const Val &a, long long i)
{
    const unsigned char *q = a.p + (size_t)i * a.tsize;
    switch (a.tkind) {
    case 'i':
        switch (a.tsize) { case 1: return *(const int8_t *)q; case 2: { int16_t v; memcpy(&v, q, 2); return v; }
                           case 4: { int32_t v; memcpy(&v, q, 4); return v; } default: { int64_t v; memcpy(&v, q, 8); return v; } }
    case 'u': case 'b':
        switch (a.tsize) { case 1: return *q; case 2: { uint16_t v; memcpy(&v, q, 2); return v; }
                           case 4: { uint32_t v; memcpy(&v, q, 4); return v; } default: { uint64_t v; memcpy(&v, q, 8); return (long long)v; } }
    default:            // 'f': numpy's astype(int32) truncates toward zero
        if (a.tsize == 4) { float v; memcpy(&v, q, 4); return (long long)v; }
        if (a.tsize == 8) { double v; memcpy(&v, q, 8); return (long long)v; }
        return 0;
    }
}

// one <video>.pkl -> counts [T], ids, boxes appended to the flat vectors
int read_pkl(const char *path, int T, std::vector<unsigned char> &image, std::vector<int32_t> &counts, std::vector<int32_t> &ids,
             std::vector<int32_t> &boxes, const Err &err)
{
    if (!read_file(path, image)) return err.fail(-5, "%s: cannot be read", path);
    static thread_local std::vector<Val> pool;
    static thread_local std::vector<int> stack, memo, kids;
    pool.clear(); stack.clear(); memo.clear(); kids.clear();
    Unpickler u{image.data(), image.size(), 0, path, err, pool, stack, memo, kids, {}};
    if (int rc = u.run()) return rc;
    if (u.stack.size() != 1 || u.pool[u.stack[0]].k != K_DICT) return err.fail(-6, "%s: the pickled object is not a dict", path);
    const Val &top = u.pool[u.stack[0]];
    int bb = -1, lb = -1;
    for (int q = 0; q + 1 < top.nkid; q += 2) {                  // later duplicates win, as in a Python dict
        if (Unpickler::streq(u.pool[u.kid(top, q)], "bb")) bb = u.kid(top, q + 1);
        if (Unpickler::streq(u.pool[u.kid(top, q)], "labels")) lb = u.kid(top, q + 1);
    }
    if (bb < 0 || lb < 0) return err.fail(-6, "%s: no \"bb\" / \"labels\" entries", path);
    const Val &B = u.pool[bb], &L = u.pool[lb];
    if (B.k != K_LIST || L.k != K_LIST) return err.fail(-6, "%s: \"bb\" / \"labels\" must be lists of per-frame arrays", path);
    if (B.nkid != T || L.nkid != T) return err.fail(-7, "%s: %d / %d frames, expected %d", path, B.nkid, L.nkid, T);
    for (int t = 0; t < T; ++t) {
        const Val &l = u.pool[u.kid(L, t)], &b = u.pool[u.kid(B, t)];
        if (l.k != K_ARRAY || b.k != K_ARRAY) return err.fail(-6, "%s: frame %d is not a numeric ndarray", path, t);
        long long n = 1, nb = 1;
        for (int q = 0; q < l.ndim; ++q) n *= l.shape[q];
        for (int q = 0; q < b.ndim; ++q) nb *= b.shape[q];
        if (nb != 4 * n) return err.fail(-6, "%s: frame %d has %lld labels but %lld box coordinates", path, t, n, nb);
        counts.push_back((int32_t)n);
        for (long long q = 0; q < n; ++q) ids.push_back((int32_t)element(l, q));
        for (long long q = 0; q < 4 * n; ++q) boxes.push_back((int32_t)element(b, q));
    }
    return 0;
}

// ---- <video>_bb.json: the snitch's list of [x, y, w, h] -------------------------------------------------------------------------
struct Json {
    const unsigned char *d;
    size_t n, pos = 0;
    void ws() { while (pos < n && (d[pos] == ' ' || d[pos] == '\n' || d[pos] == '\t' || d[pos] == '\r')) ++pos; }
    bool lit(char c) { ws(); if (pos < n && d[pos] == (unsigned char)c) { ++pos; return true; } return false; }
    // a string without decoding: returns [b, e) of its raw bytes; escapes are skipped (a key WITH escapes never equals the snitch's)
    bool str(size_t &b, size_t &e, bool &escaped)
    {
        ws();
        if (pos >= n || d[pos] != '"') return false;
        b = ++pos; escaped = false;
        while (pos < n && d[pos] != '"') { if (d[pos] == '\\') { escaped = true; ++pos; } ++pos; }
        if (pos >= n) return false;
        e = pos++;
        return true;
    }
    bool skip()      // any value
    {
        ws();
        if (pos >= n) return false;
        const unsigned char c = d[pos];
        if (c == '"') { size_t b, e; bool esc; return str(b, e, esc); }
        if (c == '{' || c == '[') {
            int depth = 0;
            while (pos < n) {
                const unsigned char x = d[pos];
                if (x == '"') { size_t b, e; bool esc; if (!str(b, e, esc)) return false; continue; }
                if (x == '{' || x == '[') ++depth;
                if (x == '}' || x == ']') { --depth; if (depth == 0) { ++pos; return true; } }
                ++pos;
            }
            return false;
        }
        while (pos < n && !strchr(",}] \n\t\r", d[pos])) ++pos;      // number / true / false / null
        return true;
    }
    bool integer(long long &v)
    {
        ws();
        const size_t s = pos;
        if (pos < n && (d[pos] == '-' || d[pos] == '+')) ++pos;
        const size_t digits = pos;
        while (pos < n && d[pos] >= '0' && d[pos] <= '9') ++pos;
        if (pos == digits) return false;
        if (pos < n && (d[pos] == '.' || d[pos] == 'e' || d[pos] == 'E')) return false;   // np.array(..., dtype=int64) of a float would truncate: refuse
        if (pos - s > 18) return false;
        v = strtoll(std::string((const char *)d + s, pos - s).c_str(), nullptr, 10);
        return true;
    }
};

int read_labels(const char *path, int T, std::vector<unsigned char> &image, float *out, const Err &err)
{
    static const char KEY[] = "small_gold_spl_metal_Spl_0";        // datasets.py:13
    if (!read_file(path, image)) return err.fail(-5, "%s: cannot be read", path);
    Json j{image.data(), image.size()};
    if (!j.lit('{')) return err.fail(-6, "%s: not a JSON object", path);
    bool found = false;
    std::vector<long long> quad((size_t)T * 4);
    if (!j.lit('}')) {
        while (true) {
            size_t b, e;
            bool esc;
            if (!j.str(b, e, esc) || !j.lit(':')) return err.fail(-6, "%s: malformed JSON near byte %zu", path, j.pos);
            if (!esc && e - b == sizeof(KEY) - 1 && memcmp(j.d + b, KEY, e - b) == 0) {
                // list[T] of [x, y, w, h]; a repeated key: the last one wins, as in json.load
                if (!j.lit('[')) return err.fail(-6, "%s: the snitch entry is not a list", path);
                int t = 0;
                if (!j.lit(']')) {
                    while (true) {
                        if (t >= T) return err.fail(-7, "%s: more than %d label frames", path, T);
                        if (!j.lit('[')) return err.fail(-6, "%s: label frame %d is not a list", path, t);
                        for (int q = 0; q < 4; ++q) {
                            if (q && !j.lit(',')) return err.fail(-6, "%s: label frame %d does not hold 4 values", path, t);
                            if (!j.integer(quad[(size_t)t * 4 + q])) return err.fail(-6, "%s: label frame %d holds a non-integer", path, t);
                        }
                        if (!j.lit(']')) return err.fail(-6, "%s: label frame %d does not hold 4 values", path, t);
                        ++t;
                        if (j.lit(',')) continue;
                        if (j.lit(']')) break;
                        return err.fail(-6, "%s: malformed JSON near byte %zu", path, j.pos);
                    }
                }
                if (t != T) return err.fail(-7, "%s: %d label frames, expected %d", path, t, T);
                found = true;
            } else if (!j.skip()) {
                return err.fail(-6, "%s: malformed JSON near byte %zu", path, j.pos);
            }
            if (j.lit(',')) continue;
            if (j.lit('}')) break;
            return err.fail(-6, "%s: malformed JSON near byte %zu", path, j.pos);
        }
    }
    if (!found) return err.fail(-6, "%s: no \"%s\" entry", path, KEY);
    for (int t = 0; t < T; ++t) {
        const long long x = quad[(size_t)t * 4], y = quad[(size_t)t * 4 + 1], w = quad[(size_t)t * 4 + 2], h = quad[(size_t)t * 4 + 3];
        out[t * 4 + 0] = (float)((double)x / 320.0);               // datasets.py:38-45, then torch.tensor(dtype=float32)
        out[t * 4 + 1] = (float)((double)y / 240.0);
        out[t * 4 + 2] = (float)((double)(x + w) / 320.0);
        out[t * 4 + 3] = (float)((double)(y + h) / 240.0);
    }
    return 0;
}

}  // namespace

// n_clips clips read from their files and encoded: pkl_paths[c] -> boxes [c][T][15][n_tracks] fp32 + index [c][T] int64 (may be
// null); json_paths (may be null, as may its entries) -> labels [c][T][4] fp32 (may be null).  Returns 0, or a negative code with
// a message naming the file in err (-5 unreadable, -6 refused / malformed, -7 not T frames, -1 / -3 bad arguments).
extern "C" __attribute__((visibility("default")))
int opnet_load_clips_f32(const char *const *pkl_paths, const char *const *json_paths, int n_clips, int T, int n_tracks,
                         const uint8_t *is_cone, int n_classes, float *boxes_out, int64_t *index_out, float *labels_out, char *err,
                         int err_len)
{
    const Err e{err, err_len};
    if (err && err_len > 0) err[0] = 0;
    if (!pkl_paths || !is_cone || !boxes_out) return e.fail(-1, "null pointer");
    if (n_clips < 0 || T <= 0 || (n_tracks != 5 && n_tracks != 6)) return e.fail(-3, "bad shape (n_clips=%d T=%d n_tracks=%d)", n_clips, T, n_tracks);
    std::vector<unsigned char> image;
    std::vector<int32_t> counts, ids, boxes;
    std::vector<int64_t> first_frame(n_clips), first_det(n_clips + 1, 0);
    counts.reserve((size_t)n_clips * T);
    for (int c = 0; c < n_clips; ++c) {
        if (!pkl_paths[c]) return e.fail(-1, "null path");
        first_frame[c] = (int64_t)counts.size();
        first_det[c] = (int64_t)ids.size();
        if (int rc = read_pkl(pkl_paths[c], T, image, counts, ids, boxes, e)) return rc;
        if (labels_out && json_paths && json_paths[c])
            if (int rc = read_labels(json_paths[c], T, image, labels_out + (size_t)c * T * 4, e)) return rc;
    }
    first_det[n_clips] = (int64_t)ids.size();
    static const int32_t none = 0;
    const int rc = opnet_encode_clips_f32(counts.data(), ids.empty() ? &none : ids.data(), boxes.empty() ? &none : boxes.data(),
                                          first_frame.data(), first_det.data(), n_clips, T, n_tracks, is_cone, n_classes, boxes_out,
                                          index_out);
    if (rc) return e.fail(rc, "opnet_encode_clips_f32 failed (code %d)", rc);
    return 0;
}

// The same over n_threads host threads (clips are independent; thread k takes every n_threads-th run of 4 clips).  This is the
// loader itself: a driver calls it for a few hundred clips at a time from a prefetch thread and gets its input tensors - in pinned
// memory if it passes pinned buffers - without worker processes, inter-process queues or a collate copy (a torch DataLoader
// with 8 workers delivered 6 k clips/s from files on the box, bound by the receiving process).
extern "C" __attribute__((visibility("default")))
int opnet_load_clips_mt_f32(const char *const *pkl_paths, const char *const *json_paths, int n_clips, int T, int n_tracks,
                            const uint8_t *is_cone, int n_classes, float *boxes_out, int64_t *index_out, float *labels_out, char *err,
                            int err_len, int n_threads)
{
    if (err && err_len > 0) err[0] = 0;
    const int RUN = 4;
    const int n_runs = (n_clips + RUN - 1) / RUN;
    if (n_threads > n_runs) n_threads = n_runs;
    if (n_threads <= 1 || n_clips <= RUN)
        return opnet_load_clips_f32(pkl_paths, json_paths, n_clips, T, n_tracks, is_cone, n_classes, boxes_out, index_out, labels_out, err,
                                    err_len);
    if (!pkl_paths || !is_cone || !boxes_out) return Err{err, err_len}.fail(-1, "null pointer");
    if (T <= 0 || (n_tracks != 5 && n_tracks != 6)) return Err{err, err_len}.fail(-3, "bad shape (T=%d n_tracks=%d)", T, n_tracks);
    std::atomic<int> next{0}, first_rc{0};
    std::vector<std::string> msgs((size_t)n_threads, std::string(512, '\0'));
    std::vector<int> rcs((size_t)n_threads, 0);
    auto work = [&](int tid) {
        while (first_rc.load(std::memory_order_relaxed) == 0) {
            const int r = next.fetch_add(1);
            if (r >= n_runs) break;
            const int lo = r * RUN, cnt = (n_clips - lo < RUN) ? n_clips - lo : RUN;
            const int rc = opnet_load_clips_f32(pkl_paths + lo, json_paths ? json_paths + lo : nullptr, cnt, T, n_tracks, is_cone, n_classes,
                                                boxes_out + (size_t)lo * T * 15 * n_tracks, index_out ? index_out + (size_t)lo * T : nullptr,
                                                labels_out ? labels_out + (size_t)lo * T * 4 : nullptr, &msgs[tid][0], 512);
            if (rc) { rcs[tid] = rc; int zero = 0; first_rc.compare_exchange_strong(zero, tid + 1); break; }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < n_threads; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto &th : pool) th.join();
    const int bad = first_rc.load();
    if (bad) return Err{err, err_len}.fail(rcs[bad - 1], "%s", msgs[bad - 1].c_str());
    return 0;
}
