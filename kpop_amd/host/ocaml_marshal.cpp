// ocaml_marshal.cpp -- see ocaml_marshal.h
#include "ocaml_marshal.h"

#include <errno.h>
#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <thread>
#include <atomic>
#include <memory>
#include <vector>

namespace kpop_host {

namespace {

enum : uint8_t {
  PREFIX_SMALL_BLOCK = 0x80, PREFIX_SMALL_INT = 0x40, PREFIX_SMALL_STRING = 0x20,
  CODE_INT8 = 0x00, CODE_INT16 = 0x01, CODE_INT32 = 0x02, CODE_INT64 = 0x03,
  CODE_SHARED8 = 0x04, CODE_SHARED16 = 0x05, CODE_SHARED32 = 0x06, CODE_SHARED64 = 0x14,
  CODE_DOUBLE_ARRAY32_LITTLE = 0x07, CODE_BLOCK32 = 0x08, CODE_STRING8 = 0x09, CODE_STRING32 = 0x0A,
  CODE_DOUBLE_BIG = 0x0B, CODE_DOUBLE_LITTLE = 0x0C, CODE_DOUBLE_ARRAY8_BIG = 0x0D, CODE_DOUBLE_ARRAY8_LITTLE = 0x0E,
  CODE_DOUBLE_ARRAY32_BIG = 0x0F, CODE_BLOCK64 = 0x13, CODE_STRING64 = 0x15, CODE_DOUBLE_ARRAY64_BIG = 0x16,
  CODE_DOUBLE_ARRAY64_LITTLE = 0x17, CODE_CUSTOM = 0x12, CODE_CUSTOM_LEN = 0x18, CODE_CUSTOM_FIXED = 0x19
};
constexpr int CAML_BA_INT32 = 6;  // runtime/caml/bigarray.h: enum caml_ba_kind
constexpr uint32_t MAGIC_SMALL = 0x8495A6BE, MAGIC_BIG = 0x8495A6BF;

// ---------------------------------------------------------------- writer
struct Writer {
  std::string buf;
  uint64_t n_obj = 0, size32 = 0, size64 = 0;
  void u8(uint8_t v) { buf.push_back((char)v); }
  void be(uint64_t v, int bytes) {
    for (int i = bytes - 1; i >= 0; --i) buf.push_back((char)((v >> (8 * i)) & 0xFF));
  }
  void block_header(uint64_t size, int tag) {
    if (size == 0) {  // atom
      u8((uint8_t)(PREFIX_SMALL_BLOCK + tag));
      return;
    }
    if (tag < 16 && size < 8) u8((uint8_t)(PREFIX_SMALL_BLOCK + tag + (size << 4)));
    else if (size < (1ull << 22)) {
      u8(CODE_BLOCK32);
      be((size << 10) | (uint64_t)tag, 4);
    } else {
      u8(CODE_BLOCK64);
      be((size << 10) | (uint64_t)tag, 8);
    }
    ++n_obj;
    size32 += 1 + size;
    size64 += 1 + size;
  }
  void string(const std::string &s) {
    const uint64_t len = s.size();
    if (len < 0x20) u8((uint8_t)(PREFIX_SMALL_STRING + len));
    else if (len < 0x100) {
      u8(CODE_STRING8);
      u8((uint8_t)len);
    } else if (len < (1ull << 32)) {
      u8(CODE_STRING32);
      be(len, 4);
    } else {
      u8(CODE_STRING64);
      be(len, 8);
    }
    buf.append(s);
    ++n_obj;
    size32 += 1 + (len + 4) / 4;
    size64 += 1 + (len + 8) / 8;
  }
  void double_array(const double *p, uint64_t n) {
    if (n == 0) {  // Float.Array.create 0 is the atom
      u8(PREFIX_SMALL_BLOCK);
      return;
    }
    if (n < 0x100) {
      u8(CODE_DOUBLE_ARRAY8_LITTLE);
      u8((uint8_t)n);
    } else if (n < (1ull << 32)) {
      u8(CODE_DOUBLE_ARRAY32_LITTLE);
      be(n, 4);
    } else {
      u8(CODE_DOUBLE_ARRAY64_LITTLE);
      be(n, 8);
    }
    buf.append(reinterpret_cast<const char *>(p), n * 8);  // x86-64: native = little endian
    ++n_obj;
    size32 += 1 + 2 * n;
    size64 += 1 + n;
  }
  void integer(int64_t v) {
    if (v >= 0 && v < 0x40) u8((uint8_t)(PREFIX_SMALL_INT + v));
    else if (v >= -128 && v < 128) {
      u8(CODE_INT8);
      be((uint64_t)v, 1);
    } else if (v >= -32768 && v < 32768) {
      u8(CODE_INT16);
      be((uint64_t)v, 2);
    } else if (v >= -(1ll << 30) && v < (1ll << 30)) {
      u8(CODE_INT32);
      be((uint64_t)v, 4);
    } else {
      u8(CODE_INT64);
      be((uint64_t)v, 8);
    }
  }
  void string_array(const std::vector<std::string> &a) {
    block_header(a.size(), 0);
    for (const std::string &s : a) string(s);
  }
  // one-dimensional int32 c_layout Bigarray (caml_ba_serialize, runtime/bigarray.c)
  void bigarray_int32(const int32_t *p, uint64_t n) {
    u8(CODE_CUSTOM_LEN);
    buf.append("_bigarr02");
    buf.push_back('\0');
    const size_t lens = buf.size();
    buf.append(12, '\0');  // sz_32, sz_64: filled below
    const size_t start = buf.size();
    be(1, 4);              // num_dims
    be(CAML_BA_INT32, 4);  // kind | layout (c_layout = 0)
    if (n < 0xFFFF) be(n, 2);
    else {
      be(0xFFFF, 2);
      be(n, 8);
    }
    if (!p) {  // the prefix alone: the caller lays the big-endian payload down itself
      (void)start;
      const uint64_t sz32 = 5 * 4, sz64 = 5 * 8;
      for (int i = 0; i < 4; ++i) buf[lens + i] = (char)((sz32 >> (8 * (3 - i))) & 0xFF);
      for (int i = 0; i < 8; ++i) buf[lens + 4 + i] = (char)((sz64 >> (8 * (7 - i))) & 0xFF);
      ++n_obj;
      size32 += 2 + ((sz32 + 3) >> 2);
      size64 += 2 + ((sz64 + 7) >> 3);
      return;
    }
    char tmp[4096];
    for (uint64_t i = 0; i < n;) {  // appended through a small buffer: no zero-fill of the destination first
      const uint64_t m = std::min<uint64_t>(1024, n - i);
      for (uint64_t j = 0; j < m; ++j) {
        const uint32_t v = (uint32_t)p[i + j];
        tmp[4 * j] = (char)(v >> 24);
        tmp[4 * j + 1] = (char)(v >> 16);
        tmp[4 * j + 2] = (char)(v >> 8);
        tmp[4 * j + 3] = (char)v;
      }
      buf.append(tmp, (size_t)m * 4);
      i += m;
    }
    (void)start;
    // heap size of struct caml_ba_array with one dimension: (4 + num_dims) words
    const uint64_t sz32 = 5 * 4, sz64 = 5 * 8;
    for (int i = 0; i < 4; ++i) buf[lens + i] = (char)((sz32 >> (8 * (3 - i))) & 0xFF);
    for (int i = 0; i < 8; ++i) buf[lens + 4 + i] = (char)((sz64 >> (8 * (7 - i))) & 0xFF);
    ++n_obj;
    size32 += 2 + ((sz32 + 3) >> 2);
    size64 += 2 + ((sz64 + 7) >> 3);
  }
  // the header of a value whose data are `data_len` bytes (buf and whatever the caller writes after it)
  std::string header(uint64_t data_len) const {
    std::string hd;
    auto hbe = [&](uint64_t v, int bytes) {
      for (int i = bytes - 1; i >= 0; --i) hd.push_back((char)((v >> (8 * i)) & 0xFF));
    };
    const uint64_t lim = 1ull << 32;
    if (data_len >= lim || size32 >= lim || size64 >= lim) {
      hbe(MAGIC_BIG, 4);
      hbe(0, 4);
      hbe(data_len, 8);
      hbe(n_obj, 8);
      hbe(size64, 8);
    } else {
      hbe(MAGIC_SMALL, 4);
      hbe(data_len, 4);
      hbe(n_obj, 4);
      hbe(size32, 4);
      hbe(size64, 4);
    }
    return hd;
  }
  void flush(FILE *f) {
    std::string hd;
    auto hbe = [&](uint64_t v, int bytes) {
      for (int i = bytes - 1; i >= 0; --i) hd.push_back((char)((v >> (8 * i)) & 0xFF));
    };
    const uint64_t lim = 1ull << 32;
    if (buf.size() >= lim || size32 >= lim || size64 >= lim) {
      hbe(MAGIC_BIG, 4);
      hbe(0, 4);
      hbe(buf.size(), 8);
      hbe(n_obj, 8);
      hbe(size64, 8);
    } else {
      hbe(MAGIC_SMALL, 4);
      hbe(buf.size(), 4);
      hbe(n_obj, 4);
      hbe(size32, 4);
      hbe(size64, 4);
    }
    if (fwrite(hd.data(), 1, hd.size(), f) != hd.size() || fwrite(buf.data(), 1, buf.size(), f) != buf.size())
      throw Error(std::string("write failed: ") + strerror(errno));
  }
};

void write_string_value(FILE *f, const std::string &s) {
  Writer w;
  w.string(s);
  w.flush(f);
}

// ---------------------------------------------------------------- reader
struct Node {
  enum Kind : uint8_t { Int, Block, String, Double, DoubleArray, Int32Array } kind;
  int tag = 0;
  uint64_t a = 0, b = 0;  // Int: a=value; Block: a=first child slot, b=count; String: a=offset,b=len; Double*: a=offset,b=count
};

// the bytes of one marshalled value; deliberately NOT value-initialised (a 4 GB archive should be touched once, by fread)
struct Bytes {
  std::vector<uint8_t, DefaultInitAlloc<uint8_t>> v;  // (large ones come 2 MB-aligned and huge-page backed: kpop_text.h)
  void alloc(size_t bytes) {
    std::vector<uint8_t, DefaultInitAlloc<uint8_t>>().swap(v);
    v.resize(bytes ? bytes : 1);
    n = bytes;
  }
  size_t n = 0;
  size_t size() const { return n; }
  uint8_t *data() { return v.data(); }
  const uint8_t &operator[](size_t i) const { return v[i]; }
};

struct Reader {
  Bytes data;
  size_t pos = 0;
  std::vector<Node> nodes;
  std::vector<uint32_t> children;  // node ids
  std::vector<uint32_t> objects;   // object table for shared references
  std::string bytes;

  uint8_t u8() {
    if (pos >= data.size()) throw Error("marshal: truncated value");
    return data[pos++];
  }
  uint64_t be(int n) {
    uint64_t v = 0;
    for (int i = 0; i < n; ++i) v = (v << 8) | u8();
    return v;
  }
  uint32_t add(const Node &n, bool object) {
    nodes.push_back(n);
    const uint32_t id = (uint32_t)(nodes.size() - 1);
    if (object) objects.push_back(id);
    return id;
  }
  uint32_t shared(uint64_t ofs) {
    if (ofs == 0 || ofs > objects.size()) throw Error("marshal: bad shared reference");
    return objects[objects.size() - ofs];
  }
  uint32_t read_doubles(uint64_t n, bool little) {
    Node nd;
    nd.kind = Node::DoubleArray;
    nd.a = pos;  // offset of the payload in `data`: decoded on extraction (copy_doubles), never copied in between
    nd.b = n;
    nd.tag = little ? 1 : 0;
    if (n > (data.size() - pos) / 8) throw Error("marshal: truncated float array");
    pos += n * 8;
    return add(nd, true);
  }
  uint32_t read_string(uint64_t len) {
    if (pos + len > data.size()) throw Error("marshal: truncated string");
    Node nd;
    nd.kind = Node::String;
    nd.a = bytes.size();
    nd.b = len;
    bytes.append(reinterpret_cast<const char *>(&data[pos]), len);
    pos += len;
    return add(nd, true);
  }
  uint32_t read_block(uint64_t size, int tag) {
    if (size > data.size() - pos) throw Error("marshal: block larger than the value");  // every field takes >= 1 byte
    Node nd;
    nd.kind = Node::Block;
    nd.tag = tag;
    nd.b = size;
    if (size == 0) return add(nd, false);  // atoms are not entered in the object table
    const uint32_t id = add(nd, true);
    std::vector<uint32_t> kids(size);
    for (uint64_t i = 0; i < size; ++i) kids[i] = read_value();
    nodes[id].a = children.size();
    children.insert(children.end(), kids.begin(), kids.end());
    return id;
  }
  // custom block: only Bigarrays of int32 occur in KPop's archives
  uint32_t read_custom(uint8_t code) {
    std::string ident;
    for (uint8_t ch; (ch = u8()) != 0;) ident.push_back((char)ch);
    if (code == CODE_CUSTOM_LEN) pos += 12;  // sz_32, sz_64
    if (code == CODE_CUSTOM_FIXED || (ident != "_bigarr02" && ident != "_bigarray"))
      throw Error("marshal: unsupported custom block '" + ident + "'");
    const uint64_t num_dims = be(4), flags = be(4);
    if (num_dims != 1) throw Error("marshal: a one-dimensional Bigarray was expected");
    if ((flags & 0xFF) != (uint64_t)CAML_BA_INT32) throw Error("marshal: an int32 Bigarray was expected");
    uint64_t n;
    if (ident == "_bigarray") n = be(4);
    else {
      n = be(2);
      if (n == 0xFFFF) n = be(8);
    }
    if (n > (data.size() - pos) / 4) throw Error("marshal: truncated Bigarray");
    Node nd;
    nd.kind = Node::Int32Array;
    nd.a = pos;  // offset of the big-endian payload in `data`
    nd.b = n;
    pos += n * 4;
    return add(nd, true);
  }
  uint32_t read_value() {
    const uint8_t c = u8();
    if (c >= PREFIX_SMALL_BLOCK) return read_block((c >> 4) & 0x7, c & 0xF);
    if (c >= PREFIX_SMALL_INT) {
      Node nd;
      nd.kind = Node::Int;
      nd.a = c & 0x3F;
      return add(nd, false);
    }
    if (c >= PREFIX_SMALL_STRING) return read_string(c & 0x1F);
    Node nd;
    nd.kind = Node::Int;
    switch (c) {
      case CODE_INT8: nd.a = (uint64_t)(int64_t)(int8_t)u8(); return add(nd, false);
      case CODE_INT16: nd.a = (uint64_t)(int64_t)(int16_t)be(2); return add(nd, false);
      case CODE_INT32: nd.a = (uint64_t)(int64_t)(int32_t)be(4); return add(nd, false);
      case CODE_INT64: nd.a = be(8); return add(nd, false);
      case CODE_SHARED8: return shared(u8());
      case CODE_SHARED16: return shared(be(2));
      case CODE_SHARED32: return shared(be(4));
      case CODE_SHARED64: return shared(be(8));
      case CODE_BLOCK32: {
        const uint64_t hd = be(4);
        return read_block(hd >> 10, (int)(hd & 0xFF));
      }
      case CODE_BLOCK64: {
        const uint64_t hd = be(8);
        return read_block(hd >> 10, (int)(hd & 0xFF));
      }
      case CODE_STRING8: return read_string(u8());
      case CODE_STRING32: return read_string(be(4));
      case CODE_STRING64: return read_string(be(8));
      case CODE_DOUBLE_LITTLE: return read_doubles(1, true);
      case CODE_DOUBLE_BIG: return read_doubles(1, false);
      case CODE_DOUBLE_ARRAY8_LITTLE: return read_doubles(u8(), true);
      case CODE_DOUBLE_ARRAY8_BIG: return read_doubles(u8(), false);
      case CODE_DOUBLE_ARRAY32_LITTLE: return read_doubles(be(4), true);
      case CODE_DOUBLE_ARRAY32_BIG: return read_doubles(be(4), false);
      case CODE_DOUBLE_ARRAY64_LITTLE: return read_doubles(be(8), true);
      case CODE_DOUBLE_ARRAY64_BIG: return read_doubles(be(8), false);
      case CODE_CUSTOM:
      case CODE_CUSTOM_LEN:
      case CODE_CUSTOM_FIXED: return read_custom(c);
      default: throw Error("marshal: unsupported code " + std::to_string((int)c) + " (closures and abstract values are not part of a KPop archive)");
    }
  }
  // payload of a DoubleArray node appended to `out`
  void copy_doubles(const Node &nd, DVec &out) const {
    const size_t at = out.size();
    out.resize(at + nd.b);
    const uint8_t *s = &data[nd.a];
    if (nd.tag) memcpy(out.data() + at, s, nd.b * 8);
    else
      for (uint64_t i = 0; i < nd.b; ++i) {
        uint64_t v = 0;
        for (int k = 0; k < 8; ++k) v = (v << 8) | s[i * 8 + k];
        memcpy(&out[at + i], &v, 8);
      }
  }
  // first `n` elements of an Int32Array node
  std::vector<int32_t> int32s(const Node &nd, uint64_t n) const {
    std::vector<int32_t> out(n);
    const uint8_t *s = &data[nd.a];
    for (uint64_t i = 0; i < n; ++i) {
      uint32_t w;
      memcpy(&w, s + 4 * i, 4);
      out[i] = (int32_t)__builtin_bswap32(w);  // the payload is big-endian
    }
    return out;
  }
  std::string str(uint32_t id) const {
    const Node &n = nodes[id];
    if (n.kind != Node::String) throw Error("marshal: string expected");
    return bytes.substr(n.a, n.b);
  }
};

// header of one marshalled value: false at clean EOF, else the length of the data that follows
bool read_header(FILE *f, uint64_t *data_len_out) {
  uint8_t hd[32];
  size_t got = fread(hd, 1, 4, f);
  if (got == 0) return false;
  if (got != 4) throw Error("marshal: truncated header");
  const uint32_t magic = ((uint32_t)hd[0] << 24) | ((uint32_t)hd[1] << 16) | ((uint32_t)hd[2] << 8) | hd[3];
  uint64_t data_len = 0;
  auto be = [&](const uint8_t *p, int n) {
    uint64_t v = 0;
    for (int i = 0; i < n; ++i) v = (v << 8) | p[i];
    return v;
  };
  if (magic == MAGIC_SMALL) {
    if (fread(hd + 4, 1, 16, f) != 16) throw Error("marshal: truncated header");
    data_len = be(hd + 4, 4);
  } else if (magic == MAGIC_BIG) {
    if (fread(hd + 4, 1, 28, f) != 28) throw Error("marshal: truncated header");
    data_len = be(hd + 8, 8);
  } else {
    throw Error("marshal: bad magic number (not an OCaml output_value stream, or a compressed one)");
  }
  {  // a corrupt length must not turn into a giant allocation: it cannot exceed what is left of a regular file
    const long here = ftell(f);
    if (here >= 0 && fseek(f, 0, SEEK_END) == 0) {
      const long end = ftell(f);
      fseek(f, here, SEEK_SET);
      if (end >= here && data_len > (uint64_t)(end - here)) throw Error("marshal: truncated value");
    }
  }
  *data_len_out = data_len;
  return true;
}

// one marshalled value: false at clean EOF
bool read_one(FILE *f, Reader &r) {
  uint64_t data_len = 0;
  if (!read_header(f, &data_len)) return false;
  r = Reader();
  r.data.alloc(data_len);
  bool have = false;
  if (data_len >= (64u << 20)) {  // a large value of a regular file: pieces fetched by the host threads (page cache -> buffer)
    const off_t here = ftello(f);
    const int fd = fileno(f);
    if (here >= 0 && fd >= 0 && lseek(fd, 0, SEEK_CUR) >= 0) {
      std::atomic<bool> ok{true};
      parallel_for(data_len, 8u << 20, [&](size_t lo, size_t hi) {
        while (lo < hi && ok) {
          const ssize_t got = pread(fd, r.data.data() + lo, hi - lo, here + (off_t)lo);
          if (got < 0 && errno == EINTR) continue;
          if (got <= 0) ok = false;
          else lo += (size_t)got;
        }
      });
      if (!ok) throw Error("marshal: truncated value");
      if (fseeko(f, here + (off_t)data_len, SEEK_SET) != 0) throw Error("marshal: cannot seek");
      have = true;
    }
  }
  if (!have && data_len && fread(r.data.data(), 1, data_len, f) != data_len) throw Error("marshal: truncated value");
  r.read_value();
  return true;
}

void strings_of(const Reader &r, uint32_t id, std::vector<std::string> &out) {
  const Node &n = r.nodes[id];
  if (n.kind != Node::Block) throw Error("marshal: string array expected");
  out.clear();
  for (uint64_t i = 0; i < n.b; ++i) out.push_back(r.str(r.children[n.a + i]));
}

}  // namespace

// The three values of a matrix archive as {everything before the row data, the row data}: the names are small and go
// through the Writer; the rows are a fixed number of bytes each (a float-array prefix and the doubles as they stand in
// memory), so their place in the file is known without building them and they can be laid down in parallel.
struct MatrixImage {
  std::string head;        // type name, archive version, the value's header, and the record up to the data block's prefix
  size_t row_prefix = 0;   // bytes in front of each row's doubles
  char prefix[9];
  size_t row_bytes = 0;    // row_prefix + 8 * cols (1 for the atom of an empty row)
  uint64_t total() const { return head.size() + rows * row_bytes; }
  uint64_t rows = 0;
};

static MatrixImage matrix_image(const std::string &type_name, const Table &t) {
  MatrixImage im;
  Writer names;  // col_names, row_names and the block prefixes
  names.block_header(3, 0);
  names.block_header(t.col_names.size(), 0);
  for (const std::string &s : t.col_names) names.string(s);
  names.block_header(t.row_names.size(), 0);
  for (const std::string &s : t.row_names) names.string(s);
  names.block_header(t.rows(), 0);
  const uint64_t nc = t.cols(), nr = t.rows();
  {  // what Writer::double_array would put in front of a row
    Writer w;
    if (nc == 0) w.u8(PREFIX_SMALL_BLOCK);
    else if (nc < 0x100) {
      w.u8(CODE_DOUBLE_ARRAY8_LITTLE);
      w.u8((uint8_t)nc);
    } else if (nc < (1ull << 32)) {
      w.u8(CODE_DOUBLE_ARRAY32_LITTLE);
      w.be(nc, 4);
    } else {
      w.u8(CODE_DOUBLE_ARRAY64_LITTLE);
      w.be(nc, 8);
    }
    im.row_prefix = w.buf.size();
    memcpy(im.prefix, w.buf.data(), w.buf.size());
  }
  im.row_bytes = im.row_prefix + 8 * nc;
  im.rows = nr;
  const uint64_t data_len = names.buf.size() + nr * im.row_bytes;
  const uint64_t n_obj = names.n_obj + (nc ? nr : 0);
  const uint64_t size32 = names.size32 + (nc ? nr * (1 + 2 * nc) : 0), size64 = names.size64 + (nc ? nr * (1 + nc) : 0);
  Writer hd;
  const uint64_t lim = 1ull << 32;
  if (data_len >= lim || size32 >= lim || size64 >= lim) {
    hd.be(MAGIC_BIG, 4);
    hd.be(0, 4);
    hd.be(data_len, 8);
    hd.be(n_obj, 8);
    hd.be(size64, 8);
  } else {
    hd.be(MAGIC_SMALL, 4);
    hd.be(data_len, 4);
    hd.be(n_obj, 4);
    hd.be(size32, 4);
    hd.be(size64, 4);
  }
  for (const std::string *sv : {&type_name, (const std::string *)nullptr}) {  // Type.to_string m.which, then archive_version
    Writer w;
    w.string(sv ? *sv : std::string(kArchiveVersion));
    Writer h2;
    h2.be(MAGIC_SMALL, 4);
    h2.be(w.buf.size(), 4);
    h2.be(w.n_obj, 4);
    h2.be(w.size32, 4);
    h2.be(w.size64, 4);
    im.head += h2.buf;
    im.head += w.buf;
  }
  im.head += hd.buf;
  im.head += names.buf;
  return im;
}

static void lay_rows(char *dst, const MatrixImage &im, const Table &t, size_t lo, size_t hi) {
  const size_t nc = t.cols();
  for (size_t r = lo; r < hi; ++r) {
    char *p = dst + (r - lo) * im.row_bytes;
    memcpy(p, im.prefix, im.row_prefix);
    if (nc) memcpy(p + im.row_prefix, t.data.data() + r * nc, nc * 8);  // x86-64: native = little endian
  }
}

void marshal_write_matrix(FILE *f, const std::string &type_name, const Table &t) {
  const MatrixImage im = matrix_image(type_name, t);
  if (fwrite(im.head.data(), 1, im.head.size(), f) != im.head.size()) throw Error(std::string("write failed: ") + strerror(errno));
  const size_t chunk_rows = std::max<size_t>(1, (8u << 20) / std::max<size_t>(1, im.row_bytes));
  std::vector<char> buf(chunk_rows * im.row_bytes);
  for (size_t r0 = 0; r0 < t.rows(); r0 += chunk_rows) {
    const size_t r1 = std::min(t.rows(), r0 + chunk_rows);
    parallel_for(r1 - r0, 4096, [&](size_t lo, size_t hi) { lay_rows(buf.data() + lo * im.row_bytes, im, t, r0 + lo, r0 + hi); });
    const size_t n = (r1 - r0) * im.row_bytes;
    if (fwrite(buf.data(), 1, n, f) != n) throw Error(std::string("write failed: ") + strerror(errno));
  }
}

// a big archive: rows laid down by the host threads into a buffer of a few MB, which one write() then hands to the
// kernel -- on this class of machine a single stream of large writes fills the page cache faster than a mapping
// written by many threads does (tools/probes/write_probe.cpp); false = small archive, nothing written
static bool write_matrix_chunked(const std::string &path, const std::string &type_name, const Table &t) {
  const MatrixImage im = matrix_image(type_name, t);
  if (im.total() < (16u << 20)) return false;
  const int fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
  if (fd < 0) throw Error("cannot open '" + path + "': " + strerror(errno));
  auto put = [&](const char *p, size_t n) {
    while (n) {
      const ssize_t w = write(fd, p, n);
      if (w < 0) {
        if (errno == EINTR) continue;
        const std::string msg = std::string("write failed: ") + strerror(errno);
        close(fd);
        throw Error(msg);
      }
      p += w;
      n -= (size_t)w;
    }
  };
  put(im.head.data(), im.head.size());
  const size_t chunk_rows = std::max<size_t>(1, (16u << 20) / std::max<size_t>(1, im.row_bytes));
  std::vector<char> buf(chunk_rows * im.row_bytes);
  for (size_t r0 = 0; r0 < t.rows(); r0 += chunk_rows) {
    const size_t r1 = std::min(t.rows(), r0 + chunk_rows);
    parallel_for(r1 - r0, 2048, [&](size_t lo, size_t hi) { lay_rows(buf.data() + lo * im.row_bytes, im, t, r0 + lo, r0 + hi); });
    put(buf.data(), (r1 - r0) * im.row_bytes);
  }
  close(fd);
  return true;
}

bool marshal_read_matrix(FILE *f, std::string *type_name, Table *t) {
  Reader r;
  if (!read_one(f, r)) return false;
  *type_name = r.str(0);
  if (!read_one(f, r)) throw Error("marshal: archive version missing");
  const std::string version = r.str(0);
  if (version != kArchiveVersion)  // lib/Matrix.ml:831-832
    throw Error("Incompatible_archive_version(\"" + *type_name + "\", \"" + version + "\", \"" + kArchiveVersion + "\")");
  if (!read_one(f, r)) throw Error("marshal: matrix missing");
  const Node &root = r.nodes[0];
  if (root.kind != Node::Block || root.b != 3) throw Error("marshal: a 3-field matrix record was expected");
  *t = Table();
  strings_of(r, r.children[root.a + 0], t->col_names);
  strings_of(r, r.children[root.a + 1], t->row_names);
  const Node &data = r.nodes[r.children[root.a + 2]];
  if (data.kind != Node::Block || data.b != t->row_names.size()) throw Error("marshal: matrix data does not match its row names");
  const size_t nc = t->col_names.size();
  t->data.reserve(t->row_names.size() * nc);
  for (uint64_t i = 0; i < data.b; ++i) {
    const Node &row = r.nodes[r.children[data.a + i]];
    const bool empty_atom = row.kind == Node::Block && row.b == 0;
    if (!(row.kind == Node::DoubleArray || empty_atom) || (empty_atom ? 0 : row.b) != nc)
      throw Error("marshal: matrix row " + std::to_string(i) + " is not a float array of the matrix width");
    if (!empty_atom) r.copy_doubles(row, t->data);
  }
  return true;
}

// ---------------------------------------------------------------- mapped reader
// A big matrix archive read the way it was laid down: the file is mapped, the names are walked by a cursor that knows
// only the codes a matrix of strings and float arrays is made of, and the rows -- every one the same prefix and the same
// number of little-endian doubles, one after the other -- are copied out by the host threads.  Anything else in the
// stream (shared references, big-endian rows, rows of another width, a record of another shape) makes try_* return
// false and the general reader takes the file from the start.
namespace {

struct Cursor {
  const uint8_t *p, *e;
  bool ok = true;
  uint8_t u8() {
    if (p >= e) {
      ok = false;
      return 0;
    }
    return *p++;
  }
  uint64_t be(int n) {
    uint64_t v = 0;
    for (int i = 0; i < n; ++i) v = (v << 8) | u8();
    return v;
  }
  bool header(uint64_t *data_len) {  // of one output_value
    if ((size_t)(e - p) < 20) return ok = false;
    const uint32_t magic = (uint32_t)be(4);
    if (magic == MAGIC_SMALL) {
      *data_len = be(4);
      be(4);
      be(4);
      be(4);
    } else if (magic == MAGIC_BIG) {
      be(4);
      *data_len = be(8);
      be(8);
      be(8);
    } else {
      return ok = false;
    }
    return ok && *data_len <= (uint64_t)(e - p);
  }
  bool string(std::string *out) {
    const uint8_t c = u8();
    uint64_t len;
    if (c >= PREFIX_SMALL_STRING && c < PREFIX_SMALL_INT) len = c - PREFIX_SMALL_STRING;
    else if (c == CODE_STRING8) len = u8();
    else if (c == CODE_STRING32) len = be(4);
    else if (c == CODE_STRING64) len = be(8);
    else return ok = false;
    if (!ok || len > (uint64_t)(e - p)) return ok = false;
    out->assign(reinterpret_cast<const char *>(p), len);
    p += len;
    return true;
  }
  bool block(uint64_t *size) {  // tag 0
    const uint8_t c = u8();
    if (c >= PREFIX_SMALL_BLOCK) {
      if ((c & 0xF) != 0) return ok = false;
      *size = (c >> 4) & 0x7;
    } else if (c == CODE_BLOCK32) {
      const uint64_t h = be(4);
      if (h & 0x3FF) return ok = false;
      *size = h >> 10;
    } else if (c == CODE_BLOCK64) {
      const uint64_t h = be(8);
      if (h & 0x3FF) return ok = false;
      *size = h >> 10;
    } else {
      return ok = false;
    }
    return ok;
  }
  bool strings(std::vector<std::string> *out) {
    uint64_t n;
    if (!block(&n) || n > (uint64_t)(e - p)) return ok = false;
    out->resize(n);
    for (uint64_t i = 0; i < n; ++i)
      if (!string(&(*out)[i])) return false;
    return true;
  }
};

struct Mapping {
  const uint8_t *p = nullptr;
  size_t n = 0;
  int fd = -1;
  ~Mapping() {
    if (p) munmap(const_cast<uint8_t *>(p), n);
    if (fd >= 0) close(fd);
  }
  bool open_file(const std::string &path) {
    if (path.compare(0, 5, "/dev/") == 0) return false;
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < (off_t)(16u << 20)) {
      close(fd);
      return false;
    }
    void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) {
      close(fd);
      return false;
    }
    this->fd = fd;
    p = reinterpret_cast<const uint8_t *>(m);
    n = (size_t)st.st_size;
    return true;
  }
};

// one {type name, version, matrix} triple at the cursor; rows copied unless `skip_rows`
bool try_matrix(const Mapping &map, Cursor &c, std::string *type_name, Table *t, bool skip_rows) {
  uint64_t len;
  std::string version;
  if (!c.header(&len) || !c.string(type_name)) return false;
  if (!c.header(&len) || !c.string(&version)) return false;
  if (version != kArchiveVersion) return false;  // the general reader words the error
  if (!c.header(&len)) return false;
  const uint8_t *value_end = c.p + len;
  Cursor v{c.p, value_end};
  uint64_t three, nr;
  *t = Table();
  if (!v.block(&three) || three != 3 || !v.strings(&t->col_names) || !v.strings(&t->row_names) || !v.block(&nr)) return false;
  if (nr != t->row_names.size()) return false;
  const uint64_t nc = t->col_names.size();
  uint8_t prefix[9];
  size_t np;
  if (nc == 0) return false;  // atoms: leave the odd cases to the general reader
  if (nc < 0x100) {
    prefix[0] = CODE_DOUBLE_ARRAY8_LITTLE;
    prefix[1] = (uint8_t)nc;
    np = 2;
  } else if (nc < (1ull << 32)) {
    prefix[0] = CODE_DOUBLE_ARRAY32_LITTLE;
    for (int i = 0; i < 4; ++i) prefix[1 + i] = (uint8_t)(nc >> (8 * (3 - i)));
    np = 5;
  } else {
    return false;
  }
  const uint64_t row_bytes = np + 8 * nc;
  if ((uint64_t)(value_end - v.p) != nr * row_bytes) return false;
  const uint8_t *rows = v.p;
  if (!skip_rows) {
    t->data.resize(nr * nc);
    std::atomic<bool> uniform{true};
    // the names were walked through the mapping; the rows come in by pread (no page fault per 4 KB), a few MB at a
    // time per thread, and lose their prefixes on the way into the matrix
    const uint64_t rows_at = (uint64_t)(rows - map.p);
    parallel_for(nr, std::max<size_t>(1, (4u << 20) / row_bytes), [&](size_t lo, size_t hi) {
      const size_t per = std::max<size_t>(1, (4u << 20) / row_bytes);
      std::vector<uint8_t> bounce(std::min(per, hi - lo) * row_bytes);
      for (size_t r0 = lo; r0 < hi; r0 += per) {
        const size_t r1 = std::min(hi, r0 + per);
        size_t want = (r1 - r0) * row_bytes, got = 0;
        while (got < want) {
          const ssize_t g = pread(map.fd, bounce.data() + got, want - got, (off_t)(rows_at + r0 * row_bytes + got));
          if (g < 0 && errno == EINTR) continue;
          if (g <= 0) {
            uniform = false;
            return;
          }
          got += (size_t)g;
        }
        for (size_t r = r0; r < r1; ++r) {
          const uint8_t *q = bounce.data() + (r - r0) * row_bytes;
          if (memcmp(q, prefix, np) != 0) {
            uniform = false;
            return;
          }
          memcpy(t->data.data() + r * nc, q + np, nc * 8);
        }
      }
    });
    if (!uniform) return false;
  }
  c.p = value_end;
  return true;
}

}  // namespace

static FILE *open_or_throw(const std::string &path, const char *mode) {
  FILE *f = fopen(path.c_str(), mode);
  if (!f) throw Error("cannot open '" + path + "': " + strerror(errno));
  return f;
}

Table read_binary_matrix(const std::string &path, const std::string &expect_type) {
  {
    Mapping m;
    if (m.open_file(path)) {
      Cursor c{m.p, m.p + m.n};
      Table t;
      std::string ty;
      if (try_matrix(m, c, &ty, &t, false) && ty == expect_type) return t;
    }
  }
  FILE *f = open_or_throw(path, "rb");
  Table t;
  std::string ty;
  try {
    if (!marshal_read_matrix(f, &ty, &t)) throw Error("'" + path + "' is empty");
  } catch (...) {
    fclose(f);
    throw;
  }
  fclose(f);
  if (ty != expect_type) throw Error("Unexpected_type(" + ty + ", " + expect_type + ")");  // lib/Matrix.ml:841-842
  return t;
}

void write_binary_matrix(const std::string &path, const std::string &type_name, const Table &t) {
  if (write_matrix_chunked(path, type_name, t)) return;
  FILE *f = open_or_throw(path, "wb");
  try {
    marshal_write_matrix(f, type_name, t);
  } catch (...) {
    fclose(f);
    throw;
  }
  fclose(f);
}

void read_binary_twister(const std::string &path, Table *twister, Table *inertia) {  // Twister.of_binary, lib/Twister.ml:232-246
  {
    Mapping m;
    if (m.open_file(path)) {
      Cursor c{m.p, m.p + m.n};
      std::string t1, t2;
      if (try_matrix(m, c, &t1, twister, false) && t1 == "KPopTwister") {
        // the inertia is a few hundred bytes: the general reader on the tail
        FILE *f = open_or_throw(path, "rb");
        bool ok = fseek(f, (long)(c.p - m.p), SEEK_SET) == 0;
        try {
          ok = ok && marshal_read_matrix(f, &t2, inertia) && t2 == "KPopInertia";
        } catch (...) {
          ok = false;
        }
        fclose(f);
        if (ok) return;
      }
    }
  }
  FILE *f = open_or_throw(path, "rb");
  std::string t1, t2;
  try {
    if (!marshal_read_matrix(f, &t1, twister) || !marshal_read_matrix(f, &t2, inertia)) throw Error("'" + path + "' is truncated");
  } catch (...) {
    fclose(f);
    throw;
  }
  fclose(f);
  if (t1 != "KPopTwister") throw Error("Unexpected_type(KPopTwister, " + t1 + ")");
  if (t2 != "KPopInertia") throw Error("Unexpected_type(KPopInertia, " + t2 + ")");
}

// the inertia half only: the twister matrix is stepped over by its declared length (-d / -s need the metric, not
// the 8-bytes-per-coefficient matrix in front of it)
void read_binary_twister_inertia(const std::string &path, Table *inertia) {
  FILE *f = open_or_throw(path, "rb");
  std::string t2;
  try {
    Reader r;
    if (!read_one(f, r)) throw Error("'" + path + "' is empty");
    const std::string t1 = r.str(0);
    if (t1 != "KPopTwister") throw Error("Unexpected_type(KPopTwister, " + t1 + ")");
    if (!read_one(f, r)) throw Error("marshal: archive version missing");
    const std::string version = r.str(0);
    if (version != kArchiveVersion) throw Error("Incompatible_archive_version(\"" + t1 + "\", \"" + version + "\", \"" + kArchiveVersion + "\")");
    uint64_t len = 0;
    if (!read_header(f, &len)) throw Error("'" + path + "' is truncated");
    if (fseek(f, (long)len, SEEK_CUR) != 0) throw Error("'" + path + "' cannot be searched (a pipe?)");
    if (!marshal_read_matrix(f, &t2, inertia)) throw Error("'" + path + "' is truncated");
  } catch (...) {
    fclose(f);
    throw;
  }
  fclose(f);
  if (t2 != "KPopInertia") throw Error("Unexpected_type(KPopInertia, " + t2 + ")");
}

void write_binary_twister(const std::string &path, const Table &twister, const Table &inertia) {  // Twister.to_binary, :222-231
  FILE *f = open_or_throw(path, "wb");
  try {
    marshal_write_matrix(f, "KPopTwister", twister);
    marshal_write_matrix(f, "KPopInertia", inertia);
  } catch (...) {
    fclose(f);
    throw;
  }
  fclose(f);
}

// ---- '.KPopCounter' ---------------------------------------------------------------------------------------------
void write_binary_counter(const std::string &path, const CounterCore &db) {  // KMerDB.to_binary, lib/KMerDB.ml:395-413
  FILE *f = open_or_throw(path, "wb");
  try {
    write_string_value(f, "KPopCounter");
    write_string_value(f, kArchiveVersion);
    Writer w;
    const size_t n_cols = db.col_names.size(), n_rows = db.row_names.size(), n_meta = db.meta_names.size();
    {  // one allocation for the whole value: the spectra dominate
      size_t names = 0;
      for (const std::string &s : db.row_names) names += s.size() + 9;
      for (const std::string &s : db.col_names) names += s.size() + 9;
      w.buf.reserve(names + n_cols * (n_rows * 4 + 64 + n_meta * 16) + 4096);
    }
    w.block_header(8, 0);
    w.integer((int64_t)n_cols);
    w.integer((int64_t)n_rows);
    w.integer((int64_t)n_meta);
    w.string_array(db.col_names);
    w.string_array(db.row_names);
    w.string_array(db.meta_names);
    w.block_header(n_cols, 0);
    for (size_t c = 0; c < n_cols; ++c) {
      std::vector<std::string> m = c < db.meta.size() ? db.meta[c] : std::vector<std::string>();
      m.resize(n_meta);
      w.string_array(m);
    }
    w.block_header(n_cols, 0);
    // The spectra are most of the archive (4 bytes x k-mers x spectra) and every one of them is the same few prefix bytes and
    // a big-endian payload of known length: so the header can be written first, and the payloads converted by the host
    // threads a group of spectra at a time, one group being written while the next is converted.
    const size_t before = w.buf.size();
    for (size_t c = 0; c < n_cols; ++c) w.bigarray_int32(nullptr, n_rows);
    const size_t prefix_len = n_cols ? (w.buf.size() - before) / n_cols : 0;
    const std::string prefix = w.buf.substr(before, prefix_len);
    w.buf.resize(before);
    const uint64_t col_bytes = prefix_len + (uint64_t)n_rows * 4;
    const std::string hd = w.header(before + (uint64_t)n_cols * col_bytes);
    if (fwrite(hd.data(), 1, hd.size(), f) != hd.size() || fwrite(w.buf.data(), 1, w.buf.size(), f) != w.buf.size())
      throw Error(std::string("write failed: ") + strerror(errno));
    const size_t group = std::max<size_t>(1, std::min<size_t>(n_cols, (96u << 20) / std::max<uint64_t>(1, col_bytes)));
    std::vector<char, DefaultInitAlloc<char>> bufs[2];
    std::thread writer;
    std::atomic<bool> failed{false};
    for (size_t c0 = 0, g = 0; c0 < n_cols; c0 += group, ++g) {
      const size_t c1 = std::min(n_cols, c0 + group);
      auto &buf = bufs[g & 1];
      if (g >= 2 && writer.joinable()) writer.join();  // (the write two groups back used this buffer; at most one write is pending)
      buf.resize((c1 - c0) * col_bytes);
      parallel_for(c1 - c0, 1, [&](size_t lo, size_t hi) {
        for (size_t c = c0 + lo; c < c0 + hi; ++c) {
          char *dst = buf.data() + (c - c0) * col_bytes;
          memcpy(dst, prefix.data(), prefix_len);
          dst += prefix_len;
          const std::vector<int32_t> &v = db.storage[c];
          const size_t have = std::min<size_t>(v.size(), n_rows);  // columns may be physically shorter than n_rows (trailing zeros)
          for (size_t i = 0; i < have; ++i) {
            const uint32_t be = __builtin_bswap32((uint32_t)v[i]);
            memcpy(dst + 4 * i, &be, 4);
          }
          if (have < n_rows) memset(dst + 4 * have, 0, (n_rows - have) * 4);
        }
      });
      if (writer.joinable()) writer.join();
      const char *data = buf.data();
      const size_t bytes = buf.size();
      writer = std::thread([f, data, bytes, &failed] {
        if (fwrite(data, 1, bytes, f) != bytes) failed = true;
      });
    }
    if (writer.joinable()) writer.join();
    if (failed) throw Error(std::string("write failed: ") + strerror(errno));
  } catch (...) {
    fclose(f);
    throw;
  }
  fclose(f);
}

CounterCore read_binary_counter(const std::string &path) {  // KMerDB.of_binary, lib/KMerDB.ml:414-430
  FILE *f = open_or_throw(path, "rb");
  CounterCore db;
  try {
    Reader r;
    if (!read_one(f, r)) throw Error("'" + path + "' is empty");
    const std::string which = r.str(0);
    if (!read_one(f, r)) throw Error("marshal: archive version missing");
    const std::string version = r.str(0);
    if (which != "KPopCounter" || version != kArchiveVersion)  // :421-422
      throw Error("Incompatible_archive_version(\"" + which + "\", \"" + version + "\")");
    if (!read_one(f, r)) throw Error("marshal: database missing");
    const Node &root = r.nodes[0];
    if (root.kind != Node::Block || root.b != 8) throw Error("marshal: the 8-field KPopCounter record was expected");
    auto field = [&](int i) { return r.children[root.a + i]; };
    auto integer = [&](int i) {
      const Node &n = r.nodes[field(i)];
      if (n.kind != Node::Int) throw Error("marshal: integer expected");
      return (uint64_t)n.a;
    };
    const uint64_t n_cols = integer(0), n_rows = integer(1), n_meta = integer(2);
    strings_of(r, field(3), db.col_names);
    strings_of(r, field(4), db.row_names);
    strings_of(r, field(5), db.meta_names);
    // containers are truncated to their exact size on output (:402-409) but tolerate buffers that are longer
    if (db.col_names.size() < n_cols || db.row_names.size() < n_rows || db.meta_names.size() < n_meta)
      throw Error("marshal: KPopCounter name tables are shorter than the declared sizes");
    db.col_names.resize(n_cols);
    db.row_names.resize(n_rows);
    db.meta_names.resize(n_meta);
    const Node &meta = r.nodes[field(6)];
    if (meta.kind != Node::Block || meta.b < n_cols) throw Error("marshal: KPopCounter metadata table is too short");
    db.meta.resize(n_cols);
    for (uint64_t c = 0; c < n_cols; ++c) {
      strings_of(r, r.children[meta.a + c], db.meta[c]);
      if (db.meta[c].size() < n_meta) throw Error("marshal: KPopCounter metadata row is too short");
      db.meta[c].resize(n_meta);
    }
    const Node &st = r.nodes[field(7)];
    if (st.kind != Node::Block || st.b < n_cols) throw Error("marshal: KPopCounter storage is too short");
    db.storage.resize(n_cols);
    for (uint64_t c = 0; c < n_cols; ++c) {
      const Node &v = r.nodes[r.children[st.a + c]];
      if (v.kind != Node::Int32Array || v.b < n_rows) throw Error("marshal: KPopCounter spectrum " + std::to_string(c) + " is not an int32 Bigarray of n_rows elements");
    }
    parallel_for(n_cols, 1, [&](size_t lo, size_t hi) {  // (the spectra are converted by the host threads)
      for (size_t c = lo; c < hi; ++c) db.storage[c] = r.int32s(r.nodes[r.children[st.a + c]], n_rows);
    });
  } catch (...) {
    fclose(f);
    throw;
  }
  fclose(f);
  return db;
}

}  // namespace kpop_host
