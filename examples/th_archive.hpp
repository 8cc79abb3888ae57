// The reference's checkpoint files, written and read WITHOUT LibTorch.
//
// evo_motion_networks/include/evo_motion_networks/saver.h:13-39: save_torch / load_torch put a module (or an optimiser) into a
// torch::serialize::OutputArchive and write `<name>.th`; PpoGaeAgent::save writes actor.th, actor_optimizer.th, critic.th,
// critic_optimizer.th (evo_motion_networks/src/agents/ppo_gae.cpp:192-204), SoftActorCriticAgent::save its eight files
// (soft_actor_critic.cpp:182-199).  Such a file is a TorchScript archive: a ZIP (stored entries) holding
//     <name>/data.pkl                     pickle (protocol 2) of the module object tree: per module a class global
//                                         `__torch__[.___torch_mangle_k] Module`, NEWOBJ, and a dict of its attributes; tensors as
//                                         torch._utils._rebuild_tensor_v2((storage persistent id), offset, size, stride, requires_grad, OrderedDict())
//     <name>/data/<k>                     raw little-endian storage of tensor k
//     <name>/code/__torch__.py, code/__torch__/___torch_mangle_k.py   one class declaration per module (attribute names and types)
//     <name>/constants.pkl, version, byteorder
// This header writes exactly that (so that the reference's load_torch / torch::jit::load / torch.jit.load accept the file) and
// reads data.pkl + data/* of files written by the reference (a small unpickler for the opcodes torch's pickler emits).
// Held to the compiled reference in both directions by tests/test_checkpoint.py (authoring container) and to torch.jit.load
// everywhere.  Own code: ZIP and pickle are public formats; nothing of LibTorch is copied.
#pragma once
#include <cstdint>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace evm_th {

struct Node {
    enum Kind { MODULE, TENSOR_F32, TENSOR_I64, STR, INT, FLOAT, BOOL, TUPLE_FF, OTHER } kind = MODULE;   // OTHER: read, not interpreted (None, lists, ...)
    // MODULE: attributes in declaration order
    std::vector<std::pair<std::string, Node>> attrs;
    // tensors
    std::vector<int64_t> shape;
    std::vector<float> f32;
    std::vector<int64_t> i64;
    bool parameter = false;   // registered with register_parameter (listed in __parameters__, requires_grad as saved)
    bool requires_grad = false;
    // scalars
    std::string s;
    int64_t i = 0;
    double f = 0.0, f2 = 0.0;
    bool b = false;

    static Node module() { return Node(); }
    static Node param(std::vector<int64_t> shape, const float *data) {
        Node n; n.kind = TENSOR_F32; n.shape = std::move(shape); n.parameter = true; n.requires_grad = true;
        size_t k = 1; for (int64_t d : n.shape) k *= (size_t) d;
        n.f32.assign(data, data + k);
        return n;
    }
    static Node tensor(std::vector<int64_t> shape, const float *data) { Node n = param(std::move(shape), data); n.parameter = false; n.requires_grad = false; return n; }
    static Node size_tensor(int64_t v) { Node n; n.kind = TENSOR_I64; n.i64 = {v}; n.parameter = true; n.requires_grad = false; return n; }   // archive.write(key, tensor(int64))
    static Node str(const std::string &v) { Node n; n.kind = STR; n.s = v; return n; }
    static Node integer(int64_t v) { Node n; n.kind = INT; n.i = v; return n; }
    static Node real(double v) { Node n; n.kind = FLOAT; n.f = v; return n; }
    static Node boolean(bool v) { Node n; n.kind = BOOL; n.b = v; return n; }
    static Node pair_ff(double a, double c) { Node n; n.kind = TUPLE_FF; n.f = a; n.f2 = c; return n; }
    Node &add(const std::string &name, Node child) { attrs.emplace_back(name, std::move(child)); return attrs.back().second; }
    const Node *find(const std::string &name) const {
        for (const auto &a : attrs) if (a.first == name) return &a.second;
        return nullptr;
    }
    size_t numel() const { size_t k = 1; for (int64_t d : shape) k *= (size_t) d; return kind == TENSOR_I64 ? (shape.empty() ? i64.size() : k) : k; }
};

// named_parameters() order: depth first, a module's own parameters in declaration order, then its children's
inline void named_parameters(const Node &m, const std::string &prefix, std::vector<std::pair<std::string, const Node *>> &out) {
    for (const auto &a : m.attrs)
        if (a.second.kind == Node::TENSOR_F32 && a.second.parameter) out.emplace_back(prefix + a.first, &a.second);
    for (const auto &a : m.attrs)
        if (a.second.kind == Node::MODULE) named_parameters(a.second, prefix + a.first + ".", out);
}

// ---- ZIP (stored entries) ------------------------------------------------------------------------------------------------
namespace detail {
inline uint32_t crc32(const uint8_t *p, size_t n) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; table[i] = c; }
        init = true;
    }
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) c = table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}
struct ZipWriter {
    std::string out;
    struct Entry { std::string name; uint32_t crc, size, offset; };
    std::vector<Entry> entries;
    static void u16(std::string &s, uint16_t v) { s.push_back((char) (v & 255)); s.push_back((char) (v >> 8)); }
    static void u32(std::string &s, uint32_t v) { for (int k = 0; k < 4; k++) s.push_back((char) ((v >> (8 * k)) & 255)); }
    void add(const std::string &name, const void *data, size_t n) {
        Entry e{name, crc32((const uint8_t *) data, n), (uint32_t) n, (uint32_t) out.size()};
        // tensor data 64-byte aligned like PyTorch's writer (readers that mmap rely on it): pad through the extra field
        const size_t header = 30 + name.size();
        size_t pad = (64 - ((out.size() + header + 4) % 64)) % 64;
        u32(out, 0x04034b50u); u16(out, 20); u16(out, 0); u16(out, 0); u16(out, 0); u16(out, 0x21);   // version, flags, stored, time, date (1980-01-01)
        u32(out, e.crc); u32(out, e.size); u32(out, e.size);
        u16(out, (uint16_t) name.size()); u16(out, (uint16_t) (4 + pad));
        out += name;
        u16(out, 0x4246); u16(out, (uint16_t) pad);    // extra field id "FB", then padding
        out.append(pad, 'Z');
        out.append((const char *) data, n);
        entries.push_back(e);
    }
    std::string finish() {
        const uint32_t cd_off = (uint32_t) out.size();
        for (const Entry &e : entries) {
            u32(out, 0x02014b50u); u16(out, 20); u16(out, 20); u16(out, 0); u16(out, 0); u16(out, 0); u16(out, 0x21);
            u32(out, e.crc); u32(out, e.size); u32(out, e.size);
            u16(out, (uint16_t) e.name.size()); u16(out, 0); u16(out, 0); u16(out, 0); u16(out, 0); u32(out, 0); u32(out, e.offset);
            out += e.name;
        }
        const uint32_t cd_size = (uint32_t) out.size() - cd_off;
        u32(out, 0x06054b50u); u16(out, 0); u16(out, 0); u16(out, (uint16_t) entries.size()); u16(out, (uint16_t) entries.size());
        u32(out, cd_size); u32(out, cd_off); u16(out, 0);
        return out;
    }
};
struct ZipReader {
    std::string buf;
    std::map<std::string, std::pair<size_t, size_t>> files;   // name -> (offset of the data, size)
    static uint16_t u16(const std::string &s, size_t o) { return (uint16_t) ((uint8_t) s[o] | ((uint8_t) s[o + 1] << 8)); }
    static uint32_t u32(const std::string &s, size_t o) { return (uint32_t) u16(s, o) | ((uint32_t) u16(s, o + 2) << 16); }
    explicit ZipReader(const std::string &path) {
        std::ifstream f(path, std::ios::binary);
        if (!f) throw std::runtime_error("Could not find " + path);
        buf.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
        if (buf.size() < 22) throw std::runtime_error(path + ": not a zip archive");
        size_t eocd = std::string::npos;
        for (size_t o = buf.size() - 22;; o--) {
            if (u32(buf, o) == 0x06054b50u) { eocd = o; break; }
            if (o == 0 || buf.size() - o > 22 + 65535) break;
        }
        if (eocd == std::string::npos) throw std::runtime_error(path + ": no zip end record");
        uint64_t n = u16(buf, eocd + 10), off = u32(buf, eocd + 16);
        if (off == 0xFFFFFFFFu || n == 0xFFFF) {   // zip64 (PyTorch writes it for big files only): locator right before the end record
            const size_t loc = eocd - 20;
            if (u32(buf, loc) != 0x07064b50u) throw std::runtime_error(path + ": zip64 locator missing");
            uint64_t e64 = 0; memcpy(&e64, buf.data() + loc + 8, 8);
            memcpy(&n, buf.data() + e64 + 32, 8); memcpy(&off, buf.data() + e64 + 48, 8);
        }
        size_t o = (size_t) off;
        for (uint64_t k = 0; k < n; k++) {
            if (u32(buf, o) != 0x02014b50u) throw std::runtime_error(path + ": bad central directory");
            const uint16_t method = u16(buf, o + 10), nl = u16(buf, o + 28), el = u16(buf, o + 30), cl = u16(buf, o + 32);
            const size_t size = u32(buf, o + 24), lho = u32(buf, o + 42);
            const std::string name = buf.substr(o + 46, nl);
            const size_t data = lho + 30 + u16(buf, lho + 26) + u16(buf, lho + 28);
            if (method == 0) files[name] = {data, size};   // (deflated entries: only code/ files, which loading does not need)
            o += 46 + nl + el + cl;
        }
    }
    bool has(const std::string &n) const { return files.count(n) != 0; }
    std::string get(const std::string &n) const {
        auto it = files.find(n);
        if (it == files.end()) throw std::runtime_error("archive entry missing: " + n);
        return buf.substr(it->second.first, it->second.second);
    }
};

// ---- pickle, writing -----------------------------------------------------------------------------------------------------
struct Pickler {
    std::string p;
    std::vector<std::string> storages;     // raw bytes of data/<k>
    int next_class = -1;                   // -1: the root `__torch__.Module`, then ___torch_mangle_0, 1, ...
    std::vector<std::pair<std::string, std::string>> code;   // (file name, source)
    void op(char c) { p.push_back(c); }
    void global(const std::string &mod, const std::string &name) { op('c'); p += mod; p += '\n'; p += name; p += '\n'; }
    void unicode(const std::string &s) { op('X'); uint32_t n = (uint32_t) s.size(); p.append((const char *) &n, 4); p += s; }
    void integer(int64_t v) {
        if (v >= 0 && v < 256) { op('K'); p.push_back((char) v); }
        else if (v >= 0 && v < 65536) { op('M'); uint16_t w = (uint16_t) v; p.append((const char *) &w, 2); }
        else if (v >= INT32_MIN && v <= INT32_MAX) { op('J'); int32_t w = (int32_t) v; p.append((const char *) &w, 4); }
        else { op((char) 0x8a); p.push_back(8); p.append((const char *) &v, 8); }   // LONG1
    }
    void real(double v) {   // BINFLOAT: big-endian IEEE double
        op('G');
        uint64_t u; memcpy(&u, &v, 8);
        for (int k = 7; k >= 0; k--) p.push_back((char) ((u >> (8 * k)) & 255));
    }
    void int_tuple(const std::vector<int64_t> &v) { op('('); for (int64_t x : v) integer(x); op('t'); }
    void tensor(const Node &t) {
        const bool f = t.kind == Node::TENSOR_F32;
        const std::string key = std::to_string(storages.size());
        storages.emplace_back(f ? std::string((const char *) t.f32.data(), t.f32.size() * 4) : std::string((const char *) t.i64.data(), t.i64.size() * 8));
        const size_t numel = f ? t.f32.size() : t.i64.size();
        std::vector<int64_t> stride(t.shape.size(), 1);
        for (int k = (int) t.shape.size() - 2; k >= 0; k--) stride[k] = stride[k + 1] * t.shape[k + 1];
        global("torch._utils", "_rebuild_tensor_v2");
        op('(');
        op('('); unicode("storage"); global("torch", f ? "FloatStorage" : "LongStorage"); unicode(key); unicode("cpu"); integer((int64_t) numel); op('t');
        op('Q');   // BINPERSID
        integer(0);
        int_tuple(t.shape);
        int_tuple(stride);
        op(t.requires_grad ? (char) 0x88 : (char) 0x89);
        global("collections", "OrderedDict"); op(')'); op('R');
        op('t');
        op('R');
    }
    static bool identifier(const std::string &s) {
        if (s.empty() || (s[0] >= '0' && s[0] <= '9')) return false;
        for (char c : s) if (!((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9') || c == '_')) return false;
        return true;
    }
    // returns the qualified class name of the module it pickled
    std::string module(const Node &m) {
        const int id = next_class++;
        const std::string qual = id < 0 ? "__torch__" : "__torch__.___torch_mangle_" + std::to_string(id);
        global(qual, "Module");
        op(')'); op((char) 0x81); op('}'); op('(');
        std::string src = "class Module(Module):\n  __parameters__ = [";
        for (const auto &a : m.attrs) if ((a.second.kind == Node::TENSOR_F32 || a.second.kind == Node::TENSOR_I64) && a.second.parameter) src += "\"" + a.first + "\", ";
        src += "]\n  __buffers__ = []\n";
        bool annotations = false;
        for (const auto &a : m.attrs) {
            unicode(a.first);
            std::string type;
            switch (a.second.kind) {
                case Node::MODULE: type = module(a.second) + ".Module"; break;
                case Node::TENSOR_F32: case Node::TENSOR_I64: tensor(a.second); type = "Tensor"; break;
                case Node::STR: unicode(a.second.s); type = "str"; break;
                case Node::INT: integer(a.second.i); type = "int"; break;
                case Node::FLOAT: real(a.second.f); type = "float"; break;
                case Node::BOOL: op(a.second.b ? (char) 0x88 : (char) 0x89); type = "bool"; break;
                case Node::TUPLE_FF: real(a.second.f); real(a.second.f2); op((char) 0x86); type = "Tuple[float, float]"; break;
                case Node::OTHER: throw std::runtime_error("archive: attribute " + a.first + " cannot be written");
            }
            if (identifier(a.first)) src += "  " + a.first + " : " + type + "\n";
            else {
                if (!annotations) { src += "  __annotations__ = []\n"; annotations = true; }
                src += "  __annotations__[\"" + a.first + "\"] = " + type + "\n";
            }
        }
        op('u'); op('b');
        code.emplace_back(id < 0 ? "code/__torch__.py" : "code/__torch__/___torch_mangle_" + std::to_string(id) + ".py", src);
        return qual;
    }
};

// ---- pickle, reading -----------------------------------------------------------------------------------------------------
struct PVal;
using P = std::shared_ptr<PVal>;
struct PVal {
    enum T { NONE, INT, FLOAT, BOOL, STR, GLOBAL, TUPLE, DICT, OBJECT, PERSID, REDUCED, MARK, LIST } t = NONE;
    int64_t i = 0; double f = 0; bool b = false; std::string s;
    std::vector<P> items;                         // TUPLE / LIST; REDUCED: [callable, args]; PERSID: [tuple]; OBJECT: [class, state]
    std::vector<std::pair<P, P>> dict;            // DICT
};
inline P mk(PVal::T t) { auto v = std::make_shared<PVal>(); v->t = t; return v; }
inline P unpickle(const std::string &d) {
    std::vector<P> st;
    std::map<uint32_t, P> memo;
    size_t o = 0;
    auto need = [&](size_t n) { if (o + n > d.size()) throw std::runtime_error("pickle: truncated"); };
    auto pop_to_mark = [&]() {
        std::vector<P> items;
        while (!st.empty() && st.back()->t != PVal::MARK) { items.insert(items.begin(), st.back()); st.pop_back(); }
        if (st.empty()) throw std::runtime_error("pickle: no mark");
        st.pop_back();
        return items;
    };
    for (;;) {
        need(1);
        const uint8_t op = (uint8_t) d[o++];
        switch (op) {
            case 0x80: need(1); o++; break;                                       // PROTO
            case '.': if (st.empty()) throw std::runtime_error("pickle: empty stack"); return st.back();
            case 'c': { auto v = mk(PVal::GLOBAL); size_t e = d.find('\n', o); size_t e2 = d.find('\n', e + 1);
                        if (e == std::string::npos || e2 == std::string::npos) throw std::runtime_error("pickle: bad global");
                        v->s = d.substr(o, e - o) + " " + d.substr(e + 1, e2 - e - 1); o = e2 + 1; st.push_back(v); break; }
            case 'q': need(1); memo[(uint8_t) d[o++]] = st.back(); break;         // BINPUT
            case 'r': { need(4); uint32_t k; memcpy(&k, d.data() + o, 4); o += 4; memo[k] = st.back(); break; }
            case 'h': need(1); st.push_back(memo.at((uint8_t) d[o++])); break;    // BINGET
            case 'j': { need(4); uint32_t k; memcpy(&k, d.data() + o, 4); o += 4; st.push_back(memo.at(k)); break; }
            case ')': st.push_back(mk(PVal::TUPLE)); break;
            case '}': st.push_back(mk(PVal::DICT)); break;
            case ']': st.push_back(mk(PVal::LIST)); break;
            case '(': st.push_back(mk(PVal::MARK)); break;
            case 'N': st.push_back(mk(PVal::NONE)); break;
            case 0x88: { auto v = mk(PVal::BOOL); v->b = true; st.push_back(v); break; }
            case 0x89: { auto v = mk(PVal::BOOL); v->b = false; st.push_back(v); break; }
            case 'X': { need(4); uint32_t n; memcpy(&n, d.data() + o, 4); o += 4; need(n); auto v = mk(PVal::STR); v->s = d.substr(o, n); o += n; st.push_back(v); break; }
            case 'K': { need(1); auto v = mk(PVal::INT); v->i = (uint8_t) d[o++]; st.push_back(v); break; }
            case 'M': { need(2); uint16_t w; memcpy(&w, d.data() + o, 2); o += 2; auto v = mk(PVal::INT); v->i = w; st.push_back(v); break; }
            case 'J': { need(4); int32_t w; memcpy(&w, d.data() + o, 4); o += 4; auto v = mk(PVal::INT); v->i = w; st.push_back(v); break; }
            case 0x8a: { need(1); const uint8_t n = (uint8_t) d[o++]; need(n); int64_t w = 0; memcpy(&w, d.data() + o, n < 8 ? n : 8);
                         if (n > 0 && n < 8 && (d[o + n - 1] & 0x80)) w |= ~0ull << (8 * n); o += n; auto v = mk(PVal::INT); v->i = w; st.push_back(v); break; }
            case 'G': { need(8); uint64_t u = 0; for (int k = 0; k < 8; k++) u = (u << 8) | (uint8_t) d[o + k]; o += 8; auto v = mk(PVal::FLOAT); memcpy(&v->f, &u, 8); st.push_back(v); break; }
            case 't': { auto v = mk(PVal::TUPLE); v->items = pop_to_mark(); st.push_back(v); break; }
            case 0x85: case 0x86: case 0x87: { const size_t n = op - 0x84; auto v = mk(PVal::TUPLE);
                         if (st.size() < n) throw std::runtime_error("pickle: short stack");
                         v->items.assign(st.end() - n, st.end()); st.resize(st.size() - n); st.push_back(v); break; }
            case 'Q': { auto v = mk(PVal::PERSID); v->items = {st.back()}; st.pop_back(); st.push_back(v); break; }
            case 0x81: { auto args = st.back(); st.pop_back(); auto cls = st.back(); st.pop_back(); auto v = mk(PVal::OBJECT); v->items = {cls, mk(PVal::NONE)}; (void) args; st.push_back(v); break; }
            case 'R': { auto args = st.back(); st.pop_back(); auto fn = st.back(); st.pop_back(); auto v = mk(PVal::REDUCED); v->items = {fn, args}; st.push_back(v); break; }
            case 'u': { auto items = pop_to_mark(); auto &dd = st.back(); for (size_t k = 0; k + 1 < items.size(); k += 2) dd->dict.emplace_back(items[k], items[k + 1]); break; }
            case 's': { auto val = st.back(); st.pop_back(); auto key = st.back(); st.pop_back(); st.back()->dict.emplace_back(key, val); break; }
            case 'e': { auto items = pop_to_mark(); for (auto &x : items) st.back()->items.push_back(x); break; }
            case 'a': { auto x = st.back(); st.pop_back(); st.back()->items.push_back(x); break; }
            case 'b': { auto state = st.back(); st.pop_back(); st.back()->items[1] = state; break; }
            default: throw std::runtime_error("pickle: opcode " + std::to_string(op) + " not handled");
        }
    }
}
inline Node to_node(const P &v, const ZipReader &z, const std::string &root) {
    Node n;
    switch (v->t) {
        case PVal::OBJECT: {
            n.kind = Node::MODULE;
            const P &state = v->items[1];
            if (state->t == PVal::DICT)
                for (const auto &kv : state->dict) n.attrs.emplace_back(kv.first->s, to_node(kv.second, z, root));
            return n;
        }
        case PVal::REDUCED: {   // torch._utils._rebuild_tensor_v2(storage, offset, size, stride, requires_grad, hooks)
            if (v->items[0]->s.find("_rebuild_tensor") == std::string::npos) throw std::runtime_error("archive: unexpected callable " + v->items[0]->s);
            const auto &a = v->items[1]->items;
            const auto &pid = a[0]->items[0]->items;   // ('storage', type, key, device, numel)
            const bool f = pid[1]->s.find("FloatStorage") != std::string::npos;
            if (!f && pid[1]->s.find("LongStorage") == std::string::npos) throw std::runtime_error("archive: storage type " + pid[1]->s + " not handled");
            const std::string raw = z.get(root + "/data/" + pid[2]->s);
            const int64_t offset = a[1]->i;
            for (const auto &d : a[2]->items) n.shape.push_back(d->i);
            size_t numel = 1; for (int64_t d : n.shape) numel *= (size_t) d;
            std::vector<int64_t> stride; for (const auto &d : a[3]->items) stride.push_back(d->i);
            int64_t expect = 1;   // contiguous tensors only (what modules and optimisers save)
            for (int k = (int) n.shape.size() - 1; k >= 0; k--) { if (n.shape[k] != 1 && stride[k] != expect) throw std::runtime_error("archive: non-contiguous tensor"); expect *= n.shape[k]; }
            n.requires_grad = a[4]->b;
            n.parameter = n.requires_grad;
            if (f) { n.kind = Node::TENSOR_F32; if (raw.size() < (offset + numel) * 4) throw std::runtime_error("archive: short storage"); n.f32.resize(numel); memcpy(n.f32.data(), raw.data() + offset * 4, numel * 4); }
            else { n.kind = Node::TENSOR_I64; if (raw.size() < (offset + numel) * 8) throw std::runtime_error("archive: short storage"); n.i64.resize(numel); memcpy(n.i64.data(), raw.data() + offset * 8, numel * 8); }
            return n;
        }
        case PVal::STR: n.kind = Node::STR; n.s = v->s; return n;
        case PVal::INT: n.kind = Node::INT; n.i = v->i; return n;
        case PVal::FLOAT: n.kind = Node::FLOAT; n.f = v->f; return n;
        case PVal::BOOL: n.kind = Node::BOOL; n.b = v->b; return n;
        case PVal::TUPLE:
            if (v->items.size() == 2 && v->items[0]->t == PVal::FLOAT && v->items[1]->t == PVal::FLOAT) { n.kind = Node::TUPLE_FF; n.f = v->items[0]->f; n.f2 = v->items[1]->f; return n; }
            n.kind = Node::OTHER; return n;
        default: n.kind = Node::OTHER; return n;   // (a scripted nn.Module carries None / list attributes: nothing a checkpoint needs)
    }
}
}  // namespace detail

// file name without directory and extension: the archive's top directory ("actor" for actor.th)
inline std::string archive_name(const std::string &path) {
    size_t a = path.find_last_of('/');
    a = a == std::string::npos ? 0 : a + 1;
    size_t b = path.find_last_of('.');
    if (b == std::string::npos || b < a) b = path.size();
    return path.substr(a, b - a);
}

inline void save(const std::string &path, const Node &root) {
    detail::Pickler pk;
    pk.p += "\x80\x02";
    pk.module(root);
    pk.op('.');
    const std::string top = archive_name(path);
    detail::ZipWriter z;
    for (size_t k = 0; k < pk.storages.size(); k++) z.add(top + "/data/" + std::to_string(k), pk.storages[k].data(), pk.storages[k].size());
    z.add(top + "/data.pkl", pk.p.data(), pk.p.size());
    for (auto it = pk.code.rbegin(); it != pk.code.rend(); ++it) z.add(top + "/" + it->first, it->second.data(), it->second.size());
    const char consts[] = "\x80\x02).";
    z.add(top + "/constants.pkl", consts, 4);
    z.add(top + "/version", "3\n", 2);
    z.add(top + "/byteorder", "little", 6);
    const std::string bytes = z.finish();
    std::ofstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("Could not find " + path.substr(0, path.find_last_of('/')));   // saver.h:17-18
    f.write(bytes.data(), (std::streamsize) bytes.size());
}

inline Node load(const std::string &path) {   // a missing file -> std::runtime_error (saver.h:33-34)
    detail::ZipReader z(path);
    std::string root;
    for (const auto &kv : z.files) {
        const size_t k = kv.first.find("/data.pkl");
        if (k != std::string::npos && k + 9 == kv.first.size()) { root = kv.first.substr(0, k); break; }
    }
    if (root.empty()) throw std::runtime_error(path + ": no data.pkl");
    return detail::to_node(detail::unpickle(z.get(root + "/data.pkl")), z, root);
}

// params of an Adam archive: per parameter of the single group, in order: its shape and — when a step was taken — (step, exp_avg, exp_avg_sq)
struct AdamParam { std::vector<int64_t> shape; bool has_state; int64_t step; std::vector<float> exp_avg, exp_avg_sq; };

// ---- the reference's modules ---------------------------------------------------------------------------------------------
// Sequential of (Linear | LayerNorm | parameter-free activation); `layout`: 'L' Linear(out, in), 'N' LayerNorm(width), '-' activation
struct LayerSpec { char kind; int64_t out, in; };
inline Node sequential(const std::vector<LayerSpec> &layers, const float *&p) {
    Node seq = Node::module();
    int idx = 0;
    for (const LayerSpec &l : layers) {
        Node m = Node::module();
        if (l.kind == 'L') { m.add("weight", Node::param({l.out, l.in}, p)); p += l.out * l.in; m.add("bias", Node::param({l.out}, p)); p += l.out; }
        else if (l.kind == 'N') { m.add("weight", Node::param({l.out}, p)); p += l.out; m.add("bias", Node::param({l.out}, p)); p += l.out; }
        seq.add(std::to_string(idx++), std::move(m));
    }
    return seq;
}
// ActorModule (evo_motion_networks/src/networks/actor.cpp:9-28) from its flat named_parameters() vector
inline Node actor_module(int64_t S, int64_t A, int64_t H, const float *flat) {
    const float *p = flat;
    Node root = Node::module();
    root.add("head", sequential({{'L', H, S}, {'-', 0, 0}, {'N', H, 0}, {'L', H, H}, {'-', 0, 0}, {'N', H, 0}}, p));
    root.add("mu", sequential({{'L', A, H}, {'-', 0, 0}}, p));
    root.add("sigma", sequential({{'L', A, H}, {'-', 0, 0}}, p));
    return root;
}
// CriticModule (critic.cpp:8-21)
inline Node critic_module(int64_t S, int64_t H, const float *flat) {
    const float *p = flat;
    Node root = Node::module();
    root.add("critic", sequential({{'L', H, S}, {'-', 0, 0}, {'N', H, 0}, {'L', H, H}, {'-', 0, 0}, {'N', H, 0}, {'L', 1, H}}, p));
    return root;
}
// QNetworkModule (q_net.cpp:8-31): Sequential of three Linear -> Mish -> LayerNorm blocks and Linear(H, 1) on [state, action]
inline Node q_module(int64_t S, int64_t A, int64_t H, const float *flat) {
    const float *p = flat;
    Node root = Node::module();
    root.add("q_network", sequential({{'L', H, S + A}, {'-', 0, 0}, {'N', H, 0}, {'L', H, H}, {'-', 0, 0}, {'N', H, 0}, {'L', H, H}, {'-', 0, 0}, {'N', H, 0}, {'L', 1, H}}, p));
    return root;
}
// EntropyParameter (entropy.cpp:7-15): one registered parameter `log_alpha` [n]
inline Node entropy_module(const float *log_alpha, int64_t n = 1) {
    Node root = Node::module();
    root.add("log_alpha", Node::param({n}, log_alpha));
    return root;
}
// the per-parameter split of a module's flat Adam moments (named_parameters() order), as torch::optim::Adam keeps them
inline std::vector<AdamParam> adam_params_of(const Node &module, int64_t step, const float *exp_avg, const float *exp_avg_sq);
// the flat named_parameters() vector of a loaded module, checked against the expected total
inline std::vector<float> flat_parameters(const Node &root, size_t expect, const std::string &what) {
    std::vector<std::pair<std::string, const Node *>> ps;
    named_parameters(root, "", ps);
    std::vector<float> out;
    for (const auto &kv : ps) out.insert(out.end(), kv.second->f32.begin(), kv.second->f32.end());
    if (out.size() != expect) throw std::runtime_error(what + ": checkpoint holds " + std::to_string(out.size()) + " parameters, the module expects " + std::to_string(expect));
    return out;
}

// ---- torch::optim::Adam archives (torch/csrc/api/include/torch/optim/serialize.h, format "1.5.0") -------------------------

inline Node adam_archive(const std::vector<AdamParam> &params, double lr, double beta1 = 0.9, double beta2 = 0.999, double eps = 1e-8,
                         double weight_decay = 0.0, bool amsgrad = false) {
    Node root = Node::module();
    root.add("pytorch_version", Node::str("1.5.0"));
    Node state = Node::module();
    for (size_t k = 0; k < params.size(); k++) {
        if (!params[k].has_state) continue;
        Node st = Node::module();
        st.add("step", Node::integer(params[k].step));
        st.add("exp_avg", Node::tensor(params[k].shape, params[k].exp_avg.data()));
        st.add("exp_avg_sq", Node::tensor(params[k].shape, params[k].exp_avg_sq.data()));
        state.add(std::to_string(1000 + k), std::move(st));     // the key: a decimal string (the saving process's address in the reference)
    }
    root.add("state", std::move(state));
    Node groups = Node::module();
    groups.add("param_groups/size", Node::size_tensor(1));
    Node g0 = Node::module();
    g0.add("params/size", Node::size_tensor((int64_t) params.size()));
    for (size_t k = 0; k < params.size(); k++) g0.add("params/" + std::to_string(k), Node::str(std::to_string(1000 + k)));
    Node opt = Node::module();
    opt.add("lr", Node::real(lr));
    opt.add("betas", Node::pair_ff(beta1, beta2));
    opt.add("eps", Node::real(eps));
    opt.add("weight_decay", Node::real(weight_decay));
    opt.add("amsgrad", Node::boolean(amsgrad));
    g0.add("options", std::move(opt));
    groups.add("param_groups/0", std::move(g0));
    root.add("param_groups", std::move(groups));
    return root;
}
// reads an Adam archive back into per-parameter state in group order (shapes come from the caller's module)
inline void adam_from_archive(const Node &root, std::vector<AdamParam> &params, double *lr_out = nullptr) {
    const Node *state = root.find("state"), *groups = root.find("param_groups");
    if (!state || !groups) throw std::runtime_error("optimizer archive: state / param_groups missing");
    const Node *g0 = groups->find("param_groups/0");
    if (!g0) throw std::runtime_error("optimizer archive: param_groups/0 missing");
    const Node *sz = g0->find("params/size");
    const size_t n = sz && !sz->i64.empty() ? (size_t) sz->i64[0] : 0;
    if (n != params.size()) throw std::runtime_error("optimizer archive: " + std::to_string(n) + " parameters, the module has " + std::to_string(params.size()));
    for (size_t k = 0; k < n; k++) {
        const Node *key = g0->find("params/" + std::to_string(k));
        const Node *st = key ? state->find(key->s) : nullptr;
        params[k].has_state = st != nullptr;
        if (!st) continue;
        const Node *step = st->find("step"), *m = st->find("exp_avg"), *v = st->find("exp_avg_sq");
        if (!step || !m || !v) throw std::runtime_error("optimizer archive: incomplete state entry");
        size_t numel = 1; for (int64_t d : params[k].shape) numel *= (size_t) d;
        if (m->f32.size() != numel || v->f32.size() != numel) throw std::runtime_error("optimizer archive: moment size mismatch");
        params[k].step = step->i; params[k].exp_avg = m->f32; params[k].exp_avg_sq = v->f32;
    }
    if (lr_out) { const Node *o = g0->find("options"); const Node *lr = o ? o->find("lr") : nullptr; if (lr) *lr_out = lr->f; }
}

inline std::vector<AdamParam> adam_params_of(const Node &module, int64_t step, const float *exp_avg, const float *exp_avg_sq) {
    std::vector<std::pair<std::string, const Node *>> ps;
    named_parameters(module, "", ps);
    std::vector<AdamParam> out;
    size_t off = 0;
    for (const auto &kv : ps) {
        AdamParam a;
        a.shape = kv.second->shape;
        const size_t n = kv.second->f32.size();
        a.has_state = step > 0;        // torch::optim::Adam creates a parameter's state at its first step
        a.step = step;
        if (a.has_state) { a.exp_avg.assign(exp_avg + off, exp_avg + off + n); a.exp_avg_sq.assign(exp_avg_sq + off, exp_avg_sq + off + n); }
        off += n;
        out.push_back(std::move(a));
    }
    return out;
}
// ... and back: flat moments (zeros where a parameter has no state) and the step count (the largest one; torch steps them together)
inline int64_t flat_adam(const std::vector<AdamParam> &params, std::vector<float> &exp_avg, std::vector<float> &exp_avg_sq) {
    int64_t step = 0;
    exp_avg.clear(); exp_avg_sq.clear();
    for (const AdamParam &a : params) {
        size_t n = 1; for (int64_t d : a.shape) n *= (size_t) d;
        if (a.has_state) { exp_avg.insert(exp_avg.end(), a.exp_avg.begin(), a.exp_avg.end()); exp_avg_sq.insert(exp_avg_sq.end(), a.exp_avg_sq.begin(), a.exp_avg_sq.end()); if (a.step > step) step = a.step; }
        else { exp_avg.insert(exp_avg.end(), n, 0.f); exp_avg_sq.insert(exp_avg_sq.end(), n, 0.f); }
    }
    return step;
}

}  // namespace evm_th
