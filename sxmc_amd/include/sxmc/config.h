// config.h -- the input layer either side of the hot path, in C++ and ROOT-free:
//
//   sxmc::load_config              FitConfig::FitConfig       src/config.cpp:19-297 (the reference's JSON schema;
//                                  Observable / Systematic / Source from JSON: observable.cpp, systematic.cpp, source.cpp)
//   sxmc::read_table               read_float_vector_ttree    src/io/ttree_io.cpp:21-159 (first TTree of a ROOT file ->
//                                  row-major float matrix + field names); here: an .npz holding one 1-D array per field
//                                  (what numpy.savez writes: a ZIP of stored .npy members; int / float / double / bool
//                                  -> float32), or a 2-D float32 .npy with the field names given by the caller
//   sxmc::read_dataset_to_samples  Signal::read_dataset_to_samples   src/signal.cpp:50-109 (cuts inclusive, column
//                                  packing [sample fields..., DATASET])
//
// The JSON reader is this file's own (objects, arrays, strings, numbers, true / false / null, // and /* */ comments
// as the reference's jsoncpp accepts them, README.md:64-65).  Objects remember their members in KEY ORDER (strcmp), as
// jsoncpp 0.6's std::map does: FitConfig walks `signals` that way when it numbers sources and systematic parameters
// (config.cpp:97-151), so the parameter order of a fit depends on it.
// The Python counterpart is sxmc_amd/io.py; tests/test_io_cpu.py (through tests/cpp/config_dump.cpp) checks the two against
// each other on the same files.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "fit_types.h"

namespace sxmc {

struct ConfigError : std::runtime_error {
  explicit ConfigError(const std::string& m) : std::runtime_error(m) {}
};

// ------------------------------------------------------------------------------------------------ JSON
namespace json {

class Value {
 public:
  enum Kind { Null, Bool, Number, String, Array, Object };
  Kind kind = Null;
  bool b = false;
  double num = 0;
  std::string str;
  std::vector<Value> items;                       // Array
  std::map<std::string, Value> members;           // Object, in key (byte) order like jsoncpp 0.6
  std::vector<std::string> file_order;            // Object: keys as written

  bool isMember(const std::string& k) const { return kind == Object && members.count(k) > 0; }
  const Value& operator[](const std::string& k) const {
    static const Value null_value;
    if (kind != Object) return null_value;
    auto it = members.find(k);
    return it == members.end() ? null_value : it->second;
  }
  const Value& operator[](size_t i) const { return items.at(i); }
  size_t size() const { return kind == Array ? items.size() : kind == Object ? members.size() : 0; }
  bool isNull() const { return kind == Null; }

  double asDouble(const char* what = "value") const {
    if (kind == Number) return num;
    if (kind == Bool) return b ? 1.0 : 0.0;
    throw ConfigError(std::string(what) + ": not a number");
  }
  float asFloat(const char* what = "value") const { return (float)asDouble(what); }
  long long asInt(const char* what = "value") const {
    const double d = asDouble(what);
    if (!(d == std::floor(d)) || !(std::fabs(d) < 9.0e18)) throw ConfigError(std::string(what) + ": not an integer");
    return (long long)d;
  }
  bool asBool(const char* what = "value") const {
    if (kind == Bool) return b;
    if (kind == Number) return num != 0;
    throw ConfigError(std::string(what) + ": not a boolean");
  }
  const std::string& asString(const char* what = "value") const {
    if (kind != String) throw ConfigError(std::string(what) + ": not a string");
    return str;
  }
  // Json::Value::get(key, default)
  double get(const std::string& k, double dflt) const { return isMember(k) ? (*this)[k].asDouble(k.c_str()) : dflt; }
  bool get(const std::string& k, bool dflt) const { return isMember(k) ? (*this)[k].asBool(k.c_str()) : dflt; }
  std::string get(const std::string& k, const char* dflt) const {
    return isMember(k) ? (*this)[k].asString(k.c_str()) : std::string(dflt);
  }
};

class Parser {
 public:
  explicit Parser(const std::string& t) : text(t) {}
  Value parse() {
    Value v = value();
    skip();
    if (pos != text.size()) fail("unexpected text after the document");
    return v;
  }

 private:
  const std::string& text;
  size_t pos = 0;
  int depth = 0;   // nesting of the value being read (a document is refused beyond 256 levels: the reader recurses)

  [[noreturn]] void fail(const std::string& msg) const {
    size_t line = 1, col = 1;
    for (size_t i = 0; i < pos && i < text.size(); i++) {
      if (text[i] == '\n') {
        line++;
        col = 1;
      } else {
        col++;
      }
    }
    throw ConfigError("JSON parse error: line " + std::to_string(line) + ", column " + std::to_string(col) + ": " + msg);
  }
  void skip() {
    for (;;) {
      while (pos < text.size() && (text[pos] == ' ' || text[pos] == '\t' || text[pos] == '\n' || text[pos] == '\r')) pos++;
      if (pos + 1 < text.size() && text[pos] == '/' && text[pos + 1] == '/') {
        while (pos < text.size() && text[pos] != '\n') pos++;
      } else if (pos + 1 < text.size() && text[pos] == '/' && text[pos + 1] == '*') {
        const size_t e = text.find("*/", pos + 2);
        if (e == std::string::npos) fail("unterminated comment");
        pos = e + 2;
      } else {
        return;
      }
    }
  }
  Value value() {
    struct Depth {
      int& d;
      explicit Depth(int& x) : d(x) { d++; }
      ~Depth() { d--; }
    } guard(depth);
    if (depth > 256) fail("nested too deeply");
    skip();
    if (pos >= text.size()) fail("unexpected end of text");
    const char c = text[pos];
    Value v;
    if (c == '{') {
      v.kind = Value::Object;
      pos++;
      skip();
      if (pos < text.size() && text[pos] == '}') {
        pos++;
        return v;
      }
      for (;;) {
        skip();
        if (pos >= text.size() || text[pos] != '"') fail("expected a member name");
        const std::string key = string();
        skip();
        if (pos >= text.size() || text[pos] != ':') fail("missing ':' after a member name");
        pos++;
        Value m = value();
        if (!v.members.count(key)) v.file_order.push_back(key);
        v.members[key] = std::move(m);
        skip();
        if (pos < text.size() && text[pos] == ',') {
          pos++;
          continue;
        }
        if (pos < text.size() && text[pos] == '}') {
          pos++;
          return v;
        }
        fail("missing ',' or '}' in object declaration");
      }
    }
    if (c == '[') {
      v.kind = Value::Array;
      pos++;
      skip();
      if (pos < text.size() && text[pos] == ']') {
        pos++;
        return v;
      }
      for (;;) {
        v.items.push_back(value());
        skip();
        if (pos < text.size() && text[pos] == ',') {
          pos++;
          continue;
        }
        if (pos < text.size() && text[pos] == ']') {
          pos++;
          return v;
        }
        fail("missing ',' or ']' in array declaration");
      }
    }
    if (c == '"') {
      v.kind = Value::String;
      v.str = string();
      return v;
    }
    if (text.compare(pos, 4, "true") == 0) {
      pos += 4;
      v.kind = Value::Bool;
      v.b = true;
      return v;
    }
    if (text.compare(pos, 5, "false") == 0) {
      pos += 5;
      v.kind = Value::Bool;
      return v;
    }
    if (text.compare(pos, 4, "null") == 0) {
      pos += 4;
      return v;
    }
    if (c == '-' || (c >= '0' && c <= '9')) {
      const char* b = text.c_str() + pos;
      char* e = nullptr;
      v.num = std::strtod(b, &e);
      if (e == b) fail("malformed number");
      pos += (size_t)(e - b);
      v.kind = Value::Number;
      return v;
    }
    fail(std::string("unexpected character '") + c + "'");
  }
  std::string string() {
    std::string out;
    pos++;   // the opening quote
    for (;;) {
      if (pos >= text.size()) fail("unterminated string");
      const char c = text[pos++];
      if (c == '"') return out;
      if (c != '\\') {
        out.push_back(c);
        continue;
      }
      if (pos >= text.size()) fail("unterminated string");
      const char e = text[pos++];
      switch (e) {
        case '"': out.push_back('"'); break;
        case '\\': out.push_back('\\'); break;
        case '/': out.push_back('/'); break;
        case 'b': out.push_back('\b'); break;
        case 'f': out.push_back('\f'); break;
        case 'n': out.push_back('\n'); break;
        case 'r': out.push_back('\r'); break;
        case 't': out.push_back('\t'); break;
        case 'u': {
          if (pos + 4 > text.size()) fail("truncated \\u escape");
          const unsigned cp = (unsigned)std::strtoul(text.substr(pos, 4).c_str(), nullptr, 16);
          pos += 4;
          if (cp < 0x80) {
            out.push_back((char)cp);
          } else if (cp < 0x800) {
            out.push_back((char)(0xC0 | (cp >> 6)));
            out.push_back((char)(0x80 | (cp & 0x3F)));
          } else {
            out.push_back((char)(0xE0 | (cp >> 12)));
            out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
            out.push_back((char)(0x80 | (cp & 0x3F)));
          }
          break;
        }
        default: fail("bad escape in string");
      }
    }
  }
};

inline Value parse(const std::string& text) { return Parser(text).parse(); }

}  // namespace json

// ------------------------------------------------------------------------------------------------ tables
namespace detail {

inline std::string read_file(const std::string& path) {
  std::ifstream f(path.c_str(), std::ios::binary);
  if (!f) throw ConfigError("cannot open " + path);
  std::ostringstream ss;
  ss << f.rdbuf();
  return ss.str();
}
inline uint16_t le16(const unsigned char* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t le32(const unsigned char* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint64_t le64(const unsigned char* p) { return (uint64_t)le32(p) | ((uint64_t)le32(p + 4) << 32); }

/** One .npy image (numpy format 1.0 - 3.0): dtype, shape, and where the data starts. */
struct NpyView {
  std::string descr;
  std::vector<size_t> shape;
  bool fortran = false;
  const unsigned char* data = nullptr;
  size_t nbytes = 0;
  size_t count() const {   // (saturates instead of wrapping: a hostile shape must not look small)
    size_t n = 1;
    for (size_t s : shape) {
      if (s != 0 && n > ((size_t)-1) / s) return (size_t)-1;
      n *= s;
    }
    return n;
  }
};

inline NpyView parse_npy(const unsigned char* p, size_t n, const std::string& what) {
  if (n < 10 || std::memcmp(p, "\x93NUMPY", 6) != 0) throw ConfigError(what + ": not an .npy image");
  const int major = p[6];
  size_t hlen, hoff;
  if (major == 1) {
    hlen = le16(p + 8);
    hoff = 10;
  } else {
    if (n < 12) throw ConfigError(what + ": truncated .npy header");
    hlen = le32(p + 8);
    hoff = 12;
  }
  if (hoff + hlen > n) throw ConfigError(what + ": truncated .npy header");
  const std::string h((const char*)p + hoff, hlen);
  NpyView v;
  auto field = [&](const char* key) -> size_t {
    const size_t k = h.find(std::string("'") + key + "'");
    const size_t c = k == std::string::npos ? k : h.find(':', k);
    if (c == std::string::npos) throw ConfigError(what + ": .npy header has no '" + key + "'");
    return c + 1;
  };
  {
    const size_t a = h.find('\'', field("descr"));
    const size_t b = a == std::string::npos ? a : h.find('\'', a + 1);
    if (a == std::string::npos || b == std::string::npos) throw ConfigError(what + ": structured dtypes are not supported");
    v.descr = h.substr(a + 1, b - a - 1);
  }
  {
    const size_t a = h.find_first_not_of(' ', field("fortran_order"));
    v.fortran = a != std::string::npos && h.compare(a, 4, "True") == 0;
  }
  {
    const size_t a = h.find('(', field("shape"));
    const size_t b = a == std::string::npos ? a : h.find(')', a);
    if (a == std::string::npos || b == std::string::npos) throw ConfigError(what + ": .npy header has no shape");
    std::string sh = h.substr(a + 1, b - a - 1);
    for (char& c : sh)
      if (c == ',') c = ' ';
    std::istringstream is(sh);
    unsigned long long d;
    size_t total = 1;
    while (is >> d) {
      // (a shape whose product does not fit the file is caught below; one that overflows size_t here)
      if (d != 0 && total > (size_t)-1 / (size_t)d) throw ConfigError(what + ": .npy shape overflows");
      total *= (size_t)d;
      v.shape.push_back((size_t)d);
    }
    if (v.shape.size() > 8) throw ConfigError(what + ": .npy shape has too many dimensions");
  }
  v.data = p + hoff + hlen;
  v.nbytes = n - hoff - hlen;
  return v;
}

/** element i of an .npy image as float32 (numpy's astype(float32)) */
inline void npy_to_float(const NpyView& v, float* out, size_t stride, const std::string& what) {
  const std::string& d = v.descr;
  if (d.size() < 3 || (d[0] != '<' && d[0] != '|' && d[0] != '=')) {
    throw ConfigError(what + ": dtype " + d + " is not supported (little-endian int / float / double / bool only)");
  }
  const char kind = d[1];
  const int width = std::atoi(d.c_str() + 2);
  if (width != 1 && width != 2 && width != 4 && width != 8) {
    throw ConfigError(what + ": dtype " + d + " is not supported (int / float / double / bool only)");
  }
  const size_t n = v.count();
  if (n > v.nbytes / (size_t)width) throw ConfigError(what + ": truncated data");
  const unsigned char* p = v.data;
  for (size_t i = 0; i < n; i++, p += width) {
    float f;
    if (kind == 'f' && width == 4) {
      std::memcpy(&f, p, 4);
    } else if (kind == 'f' && width == 8) {
      double x;
      std::memcpy(&x, p, 8);
      f = (float)x;
    } else if (kind == 'i' && width == 1) {
      f = (float)(int8_t)p[0];
    } else if (kind == 'i' && width == 2) {
      f = (float)(int16_t)le16(p);
    } else if (kind == 'i' && width == 4) {
      f = (float)(int32_t)le32(p);
    } else if (kind == 'i' && width == 8) {
      f = (float)(int64_t)le64(p);
    } else if ((kind == 'u' || kind == 'b') && width == 1) {
      f = (float)p[0];
    } else if (kind == 'u' && width == 2) {
      f = (float)le16(p);
    } else if (kind == 'u' && width == 4) {
      f = (float)le32(p);
    } else if (kind == 'u' && width == 8) {
      f = (float)le64(p);
    } else {
      throw ConfigError(what + ": dtype " + d + " is not supported (int / float / double / bool only)");
    }
    out[i * stride] = f;
  }
}

struct ZipMember {
  std::string name;
  size_t offset = 0, size = 0;   // of the stored data
};

/** Members of a ZIP archive whose entries are STORED (numpy.savez; savez_compressed is refused). */
inline std::vector<ZipMember> zip_members(const unsigned char* z, size_t n, const std::string& what) {
  if (n < 22) throw ConfigError(what + ": not a ZIP archive");
  size_t eocd = std::string::npos;
  for (size_t i = n - 22 + 1; i-- > 0;) {
    if (le32(z + i) == 0x06054b50u) {
      eocd = i;
      break;
    }
    if (n - i > 22 + 65535) break;
  }
  if (eocd == std::string::npos) throw ConfigError(what + ": no ZIP end-of-central-directory record");
  uint64_t count = le16(z + eocd + 10), cd_off = le32(z + eocd + 16);
  if ((count == 0xFFFF || cd_off == 0xFFFFFFFFu) && eocd >= 20 && le32(z + eocd - 20) == 0x07064b50u) {
    const uint64_t z64 = le64(z + eocd - 20 + 8);   // zip64 end-of-central-directory record
    if (z64 > n || n - z64 < 56 || le32(z + z64) != 0x06064b50u) throw ConfigError(what + ": bad zip64 record");
    count = le64(z + z64 + 32);
    cd_off = le64(z + z64 + 48);
  }
  std::vector<ZipMember> out;
  if (cd_off > n) throw ConfigError(what + ": bad ZIP central directory offset");
  size_t p = (size_t)cd_off;
  for (uint64_t k = 0; k < count; k++) {
    if (n - p < 46 || le32(z + p) != 0x02014b50u) throw ConfigError(what + ": bad ZIP central directory");
    const unsigned method = le16(z + p + 10);
    uint64_t csize = le32(z + p + 20), usize = le32(z + p + 24), lho = le32(z + p + 42);
    const size_t nlen = le16(z + p + 28), xlen = le16(z + p + 30), clen = le16(z + p + 32);
    if (p + 46 + nlen + xlen + clen > n) throw ConfigError(what + ": ZIP central directory entry runs past the end of the file");
    ZipMember m;
    m.name.assign((const char*)z + p + 46, nlen);
    // zip64 extra field: the values that did not fit, in this order
    const size_t xend = p + 46 + nlen + xlen;
    for (size_t x = p + 46 + nlen; x + 4 <= xend;) {
      const unsigned id = le16(z + x), len = le16(z + x + 2);
      if (x + 4 + len > xend) throw ConfigError(what + ": bad ZIP extra field");
      if (id == 0x0001) {
        size_t q = x + 4;
        const size_t qend = x + 4 + len;
        auto take = [&](uint64_t& v) {
          if (q + 8 > qend) throw ConfigError(what + ": truncated zip64 extra field");
          v = le64(z + q);
          q += 8;
        };
        if (usize == 0xFFFFFFFFu) take(usize);
        if (csize == 0xFFFFFFFFu) take(csize);
        if (lho == 0xFFFFFFFFu) take(lho);
      }
      x += 4 + len;
    }
    if (method != 0) throw ConfigError(what + ": member " + m.name + " is compressed (write it with numpy.savez, not savez_compressed)");
    if (lho > n || n - lho < 30 || le32(z + lho) != 0x04034b50u) throw ConfigError(what + ": bad ZIP local header");
    m.offset = (size_t)lho + 30 + le16(z + lho + 26) + le16(z + lho + 28);
    m.size = (size_t)usize;
    if (m.offset > n || m.size > n - m.offset) throw ConfigError(what + ": member " + m.name + " runs past the end of the file");
    (void)csize;
    out.push_back(m);
    p += 46 + nlen + xlen + clen;
  }
  return out;
}

}  // namespace detail

/** read_float_vector_ttree's role (io/ttree_io.cpp:21-159): the table of `path` as a row-major float matrix
 *  [n][fields.size()] and its field names.
 *   *.npz: one 1-D array per field, fields in the archive's order (numpy.savez(path, energy=..., radius=...));
 *   *.npy: one 2-D C-ordered array [n][nfields]; the names come from `npy_fields` (the file carries none). */
inline void read_table(const std::string& path, std::vector<float>& matrix, std::vector<std::string>& fields,
                       const std::vector<std::string>& npy_fields = {}) {
  const std::string image = detail::read_file(path);
  const unsigned char* z = (const unsigned char*)image.data();
  matrix.clear();
  fields.clear();
  if (image.size() >= 6 && std::memcmp(z, "\x93NUMPY", 6) == 0) {
    detail::NpyView v = detail::parse_npy(z, image.size(), path);
    if (v.shape.size() != 2 || v.fortran) throw ConfigError(path + ": a 2-D C-ordered array [rows][fields] is expected");
    if (npy_fields.size() != v.shape[1]) {
      throw ConfigError(path + ": " + std::to_string(v.shape[1]) + " columns but " + std::to_string(npy_fields.size()) +
                        " field names (\"fields\": [...] in the signal's configuration)");
    }
    if (v.count() > v.nbytes) throw ConfigError(path + ": the header's shape does not fit the file (truncated data)");
    matrix.resize(v.count());
    detail::npy_to_float(v, matrix.data(), 1, path);
    fields = npy_fields;
    return;
  }
  const std::vector<detail::ZipMember> members = detail::zip_members(z, image.size(), path);
  size_t nrows = 0;
  for (size_t k = 0; k < members.size(); k++) {
    const detail::ZipMember& m = members[k];
    std::string name = m.name;
    if (name.size() > 4 && name.compare(name.size() - 4, 4, ".npy") == 0) name.resize(name.size() - 4);
    detail::NpyView v = detail::parse_npy(z + m.offset, m.size, path + ":" + name);
    if (v.shape.size() != 1) {
      throw ConfigError("field '" + name + "' of " + path + ": only 1-D int/float/double/bool branches are supported");
    }
    if (k == 0) {
      nrows = v.shape[0];
      // (the header's shape is checked against the member's bytes BEFORE anything is sized by it: a corrupt header must
      //  end in a ConfigError, not in a bad_alloc or a multi-GB allocation)
      if (nrows > m.size) throw ConfigError("field '" + name + "' of " + path + ": " + std::to_string(nrows) +
                                            " rows do not fit the member's " + std::to_string(m.size) + " bytes");
      if (members.size() != 0 && nrows > ((size_t)-1) / sizeof(float) / members.size()) {
        throw ConfigError(path + ": rows x fields overflows");
      }
      matrix.assign(nrows * members.size(), 0.0f);
    } else if (v.shape[0] != nrows) {
      throw ConfigError("fields of " + path + " differ in length");
    }
    detail::npy_to_float(v, matrix.data() + k, members.size(), path + ":" + name);
    fields.push_back(name);
  }
}

/** Signal::read_dataset_to_samples (signal.cpp:50-109).  dataset: row-major [n][dataset_fields.size()];
 *  sample_fields ends with "DATASET"; an event is dropped when a field that has a cut is < lower or > upper (bounds
 *  inclusive; when several cuts name one field the LAST one counts, as in the reference's lookup table).
 *  Returns rows of sample_fields.size() floats. */
inline std::vector<float> read_dataset_to_samples(const std::vector<float>& dataset,
                                                  const std::vector<std::string>& dataset_fields, unsigned dataset_id,
                                                  const std::vector<std::string>& sample_fields,
                                                  const std::vector<Observable>& cuts,
                                                  size_t required = (size_t)-1) {
  const size_t nf = dataset_fields.size(), ns = sample_fields.size();
  if (ns == 0 || nf == 0) return {};
  std::vector<char> has_cut(nf, 0);
  std::vector<double> lo(nf, 0.0), hi(nf, 0.0);
  for (size_t i = 0; i < nf; i++)
    for (const Observable& c : cuts)
      if (c.field == dataset_fields[i]) {
        has_cut[i] = 1;
        lo[i] = c.lower;
        hi[i] = c.upper;
      }
  // `required`: how many of the leading sample fields the data set must carry (default: all -- the MC tables).  Real
  // data has no Monte Carlo truth branch; the reference maps such a field past the row (signal.cpp:72-77) and never
  // looks at it again (GetSamples keeps the observables only), so for data sets -- required = the observables -- a
  // missing non-observable field becomes a column of zeros instead of an error.
  std::vector<size_t> map;
  for (size_t i = 0; i + 1 < ns; i++) {
    const size_t idx = std::find(dataset_fields.begin(), dataset_fields.end(), sample_fields[i]) - dataset_fields.begin();
    if (idx >= nf && i < required) {   // (the reference reads past the row here)
      throw ConfigError("sample field '" + sample_fields[i] + "' is not a field of the data set");
    }
    map.push_back(idx);
  }
  const size_t n = dataset.size() / nf;
  std::vector<float> out;
  out.reserve(n * ns);
  for (size_t r = 0; r < n; r++) {
    const float* row = &dataset[r * nf];
    bool valid = true;
    for (size_t j = 0; j < nf && valid; j++) {
      if (has_cut[j] && ((double)row[j] < lo[j] || (double)row[j] > hi[j])) valid = false;
    }
    if (!valid) continue;
    for (size_t j = 0; j + 1 < ns; j++) out.push_back(map[j] < nf ? row[map[j]] : 0.0f);
    out.push_back((float)dataset_id);
  }
  return out;
}

// ------------------------------------------------------------------------------------------------ chain output
namespace detail {
inline uint32_t crc32(const unsigned char* p, size_t n) {   // (ZIP's CRC-32, reflected 0xEDB88320)
  static uint32_t table[256];
  static bool ready = false;
  if (!ready) {
    for (uint32_t i = 0; i < 256; i++) {
      uint32_t c = i;
      for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      table[i] = c;
    }
    ready = true;
  }
  uint32_t c = 0xFFFFFFFFu;
  for (size_t i = 0; i < n; i++) c = table[(c ^ p[i]) & 0xFFu] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}
inline void put16(std::string& o, unsigned v) {
  o.push_back((char)(v & 0xFF));
  o.push_back((char)((v >> 8) & 0xFF));
}
inline void put32(std::string& o, uint32_t v) {
  put16(o, v & 0xFFFFu);
  put16(o, v >> 16);
}
}  // namespace detail

/** The "ls" ntuple of one experiment (sxmc.cpp:130-141, mcmc.cpp:100-114: one column per parameter + "likelihood"),
 *  ROOT-free: an .npz with one 1-D float32 array per column -- what numpy.load and sxmc_amd/io.py read
 *  (`write_chain` there writes the same).  names.size() columns, rows.size() / names.size() rows, row-major. */
inline void write_chain_npz(const std::string& path, const std::vector<std::string>& names, const std::vector<float>& rows) {
  const size_t ncol = names.size(), nrow = ncol ? rows.size() / ncol : 0;
  if (ncol == 0 || nrow * ncol != rows.size()) throw ConfigError("write_chain_npz: rows are not a multiple of the columns");
  std::string zip, central;
  for (size_t c = 0; c < ncol; c++) {
    // one .npy image per column: magic, version 1.0, header padded so that the data starts on a multiple of 64
    std::string dict = "{'descr': '<f4', 'fortran_order': False, 'shape': (" + std::to_string(nrow) + ",), }";
    while ((10 + dict.size() + 1) % 64 != 0) dict.push_back(' ');
    dict.push_back('\n');
    std::string npy = "\x93NUMPY";
    npy.push_back('\x01');
    npy.push_back('\x00');
    detail::put16(npy, (unsigned)dict.size());
    npy += dict;
    const size_t at = npy.size();
    npy.resize(at + 4 * nrow);
    for (size_t r = 0; r < nrow; r++) std::memcpy(&npy[at + 4 * r], &rows[r * ncol + c], 4);
    if (npy.size() > 0xFFFFFFF0ull) throw ConfigError("write_chain_npz: a column beyond 4 GB");
    const std::string fname = names[c] + ".npy";
    const uint32_t crc = detail::crc32((const unsigned char*)npy.data(), npy.size()), size = (uint32_t)npy.size();
    const uint32_t offset = (uint32_t)zip.size();
    auto header = [&](std::string& o, bool is_central) {
      detail::put32(o, is_central ? 0x02014b50u : 0x04034b50u);
      if (is_central) detail::put16(o, 20);   // version made by
      detail::put16(o, 20);                    // version needed
      detail::put16(o, 0);                     // flags
      detail::put16(o, 0);                     // method: stored
      detail::put16(o, 0);                     // time
      detail::put16(o, 0x21);                  // date: 1980-01-01
      detail::put32(o, crc);
      detail::put32(o, size);
      detail::put32(o, size);
      detail::put16(o, (unsigned)fname.size());
      detail::put16(o, 0);                     // extra length
      if (is_central) {
        detail::put16(o, 0);                   // comment length
        detail::put16(o, 0);                   // disk number
        detail::put16(o, 0);                   // internal attributes
        detail::put32(o, 0);                   // external attributes
        detail::put32(o, offset);
      }
      o += fname;
    };
    header(zip, false);
    zip += npy;
    header(central, true);
    if (zip.size() > 0xFFFFFFF0ull) throw ConfigError("write_chain_npz: archive beyond 4 GB");
  }
  const uint32_t cd_off = (uint32_t)zip.size(), cd_size = (uint32_t)central.size();
  zip += central;
  detail::put32(zip, 0x06054b50u);
  detail::put16(zip, 0);
  detail::put16(zip, 0);
  detail::put16(zip, (unsigned)ncol);
  detail::put16(zip, (unsigned)ncol);
  detail::put32(zip, cd_size);
  detail::put32(zip, cd_off);
  detail::put16(zip, 0);
  std::ofstream f(path.c_str(), std::ios::binary);
  if (!f) throw ConfigError("cannot write " + path);
  f.write(zip.data(), (std::streamsize)zip.size());
  if (!f) throw ConfigError("short write to " + path);
}

// ------------------------------------------------------------------------------------------------ FitConfig
/** What FitConfig::FitConfig (config.cpp:19-297) extracts, with every signal's table loaded and cut
 *  (Signal::Signal, signal.cpp:11-47): all that build_pdfz / ensemble / ensemble_multi_gpu take. */
struct FitConfig {
  unsigned nexperiments = 0, nsteps = 0;
  std::string error_type = "contour";   //!< "contour" or "projection"
  float burnin_fraction = 0.1f;
  bool debug_mode = false;
  std::string output_prefix = "lspace";
  long long seed = 0;
  float confidence = 0.683f;
  std::string signal_name;
  std::string samples;   //!< fit.samples: a saved chain to take the intervals from INSTEAD of walking (sxmc.cpp:84-94)
  std::vector<Observable> observables, cuts;
  std::vector<Systematic> systematics;            //!< union over the signals, parameters numbered in that order
  std::vector<Source> sources;
  std::vector<std::string> sample_fields;         //!< observables, extra truth fields, "DATASET"
  size_t nfields = 0;                             //!< sample_fields.size()
  std::vector<Signal> signals;                    //!< in fit.signals order; nexpected and n_mc (BEFORE cuts) set
  std::vector<std::vector<float>> tables;         //!< tables[j]: signal j's samples, rows of nfields floats, cuts applied
  /** data sets (config.cpp:260-296): per data set id one table PER LISTED FILE, rows of nobservables + 1 floats
   *  (observables, dataset id) -- experiment i fits file i of every data set (sxmc.cpp:71-80), see experiment_data */
  std::map<unsigned, std::vector<std::vector<float>>> data;
  std::string base_dir;
  size_t rows_total() const {
    size_t n = 0;
    for (const std::vector<float>& t : tables) n += nfields ? t.size() / nfields : 0;
    return n;
  }
};

namespace detail {
inline Observable observable_from_json(const std::string& name, const json::Value& c) {   // observable.cpp
  for (const char* k : {"field", "bins", "min", "max"})
    if (!c.isMember(k)) throw ConfigError("observable '" + name + "': missing \"" + k + "\"");
  Observable o;
  o.name = name;
  o.field = c["field"].asString("field");
  o.bins = (size_t)c["bins"].asInt("bins");
  o.lower = c["min"].asFloat("min");
  o.upper = c["max"].asFloat("max");
  return o;
}
inline Systematic systematic_from_json(const std::string& name, const json::Value& c) {   // systematic.cpp
  for (const char* k : {"observable_field", "type", "mean"})
    if (!c.isMember(k)) throw ConfigError("systematic '" + name + "': missing \"" + k + "\"");
  Systematic s;
  s.name = name;
  s.title = c.get("title", "");
  s.observable_field = c["observable_field"].asString("observable_field");
  const std::string t = c["type"].asString("type");
  if (t == "shift") s.type = pdfz::Systematic::SHIFT;
  else if (t == "scale") s.type = pdfz::Systematic::SCALE;
  else if (t == "ctscale") s.type = pdfz::Systematic::CTSCALE;
  else if (t == "resolution_scale") {
    s.type = pdfz::Systematic::RESOLUTION_SCALE;
    if (!c.isMember("truth_field")) throw ConfigError("systematic '" + name + "': resolution_scale needs \"truth_field\"");
    s.truth_field = c["truth_field"].asString("truth_field");
  } else {
    throw ConfigError("Unknown systematic type " + t);
  }
  const json::Value& mean = c["mean"];
  s.npars = mean.size();
  for (size_t j = 0; j < s.npars; j++) s.means.push_back(mean[j].asDouble("mean"));
  if (c.isMember("sigma")) {
    if (c["sigma"].size() != s.npars) throw ConfigError("systematic '" + name + "': \"sigma\" and \"mean\" differ in length");
    for (size_t j = 0; j < s.npars; j++) s.sigmas.push_back(c["sigma"][j].asDouble("sigma"));
  } else {
    s.sigmas.assign(s.npars, 0.0);
  }
  s.fixed = c.get("fixed", false);
  return s;
}
template <typename T>
size_t index_with_append(std::vector<T>& v, const T& x) {   // utils.h get_index_with_append
  const size_t i = std::find(v.begin(), v.end(), x) - v.begin();
  if (i == v.size()) v.push_back(x);
  return i;
}
inline std::string join_path(const std::string& dir, const std::string& f) {
  if (f.empty() || f[0] == '/' || dir.empty()) return f;
  return dir + "/" + f;
}
}  // namespace detail

/** Parse a configuration text (the reference's schema).  load_tables: also read every signal's and data set's
 *  table relative to base_dir. */
inline FitConfig parse_config(const std::string& text, const std::string& base_dir, bool load_tables = true) {
  const json::Value root = json::parse(text);
  const json::Value& fit = root["fit"];
  const json::Value& obs_params = root["pdfs"]["observables"];
  const json::Value& sys_params = root["pdfs"]["systematics"];
  const json::Value& sig_params = root["signals"];
  const json::Value& src_params = root["sources"];
  if (fit.kind != json::Value::Object) throw ConfigError("missing \"fit\" section");
  FitConfig fc;
  fc.base_dir = base_dir;

  // general fit parameters (config.cpp:42-75)
  if (!fit.isMember("nexperiments") || !fit.isMember("nsteps")) throw ConfigError("fit: \"nexperiments\" and \"nsteps\" are required");
  const long long nexp = fit["nexperiments"].asInt("nexperiments"), nst = fit["nsteps"].asInt("nsteps");
  if (nexp <= 0 || nst <= 0) throw ConfigError("fit: nexperiments and nsteps must be positive");
  fc.nexperiments = (unsigned)nexp;
  fc.nsteps = (unsigned)nst;
  fc.error_type = fit.get("error_type", "contour");
  if (fc.error_type != "contour" && fc.error_type != "projection") {
    throw ConfigError("FitConfig: Unknown error type \"" + fc.error_type + "\"");
  }
  fc.burnin_fraction = (float)fit.get("burnin_fraction", 0.1);
  fc.debug_mode = fit.get("debug_mode", false);
  fc.output_prefix = fit.get("output_prefix", "lspace");
  fc.seed = fit.isMember("seed") ? fit["seed"].asInt("seed") : 0;
  fc.confidence = (float)fit.get("confidence", 0.683);
  fc.signal_name = fit.get("signal_name", "");
  fc.samples = fit.get("samples", "");   // config.cpp:51

  // observables and cuts (config.cpp:77-95)
  for (const json::Value& v : fit["observables"].items) {
    const std::string& name = v.asString("fit.observables[]");
    if (!obs_params.isMember(name)) throw ConfigError("observable '" + name + "' is not described in pdfs.observables");
    fc.observables.push_back(detail::observable_from_json(name, obs_params[name]));
  }
  for (const json::Value& v : fit["cuts"].items) {
    const std::string& name = v.asString("fit.cuts[]");
    if (!obs_params.isMember(name)) throw ConfigError("cut '" + name + "' is not described in pdfs.observables");
    for (const Observable& o : fc.observables)
      if (o.name == name) throw ConfigError("'" + name + "' is both an observable and a cut");
    fc.cuts.push_back(detail::observable_from_json(name, obs_params[name]));
  }

  // systematics and sources: union over the signals, walked in KEY order (config.cpp:97-151)
  short sidx = 0, pidx = 0;
  for (const auto& kv : sig_params.members) {
    const std::string& signal_name = kv.first;
    const json::Value& sc = kv.second;
    for (const json::Value& sv : sc["systematics"].items) {
      const std::string& sys_name = sv.asString("systematics[]");
      if (!sys_params.isMember(sys_name)) throw ConfigError("systematic '" + sys_name + "' is not described in pdfs.systematics");
      bool exists = false;
      for (const Systematic& s : fc.systematics) exists = exists || s.name == sys_name;
      if (!exists) {
        Systematic s = detail::systematic_from_json(sys_name, sys_params[sys_name]);
        for (size_t k = 0; k < s.npars; k++) s.pidx.push_back(pidx++);
        fc.systematics.push_back(s);
      }
    }
    if (sc.isMember("source")) {
      const std::string& src_name = sc["source"].asString("source");
      if (!src_params.isMember(src_name)) throw ConfigError("source '" + src_name + "' is not described in sources");
      bool exists = false;
      for (const Source& s : fc.sources) exists = exists || s.name == src_name;
      if (!exists) {
        const json::Value& p = src_params[src_name];   // source.cpp
        fc.sources.push_back(Source(src_name, (size_t)sidx++, (float)p.get("mean", 1.0), (float)p.get("sigma", 0.0),
                                    p.get("fixed", false)));
      }
    } else {   // the signal is a source for itself
      fc.sources.push_back(Source(signal_name, (size_t)sidx++, (float)sc.get("mean", 1.0), (float)sc.get("sigma", 0.0),
                                  sc.get("fixed", false)));
    }
  }

  // order of the sampled fields: observables, extra truth fields, DATASET (config.cpp:153-194)
  for (Observable& o : fc.observables) o.field_index = detail::index_with_append(fc.sample_fields, o.field);
  for (Systematic& s : fc.systematics) {
    const size_t index = std::find(fc.sample_fields.begin(), fc.sample_fields.end(), s.observable_field) - fc.sample_fields.begin();
    if (index >= fc.sample_fields.size()) {
      throw ConfigError("systematic '" + s.name + "': observable_field '" + s.observable_field + "' is not an observable of the fit");
    }
    s.observable_field_index = index;
    if (s.type == pdfz::Systematic::RESOLUTION_SCALE) {
      s.truth_field_index = detail::index_with_append(fc.sample_fields, s.truth_field);
    }
  }
  fc.sample_fields.push_back("DATASET");
  fc.nfields = fc.sample_fields.size();

  // signals (config.cpp:196-258, signal.cpp:11-47)
  for (const json::Value& v : fit["signals"].items) {
    const std::string& name = v.asString("fit.signals[]");
    if (!sig_params.isMember(name)) throw ConfigError("signal '" + name + "' is not described in signals");
    const json::Value& c = sig_params[name];
    for (const char* k : {"dataset", "filename"})
      if (!c.isMember(k)) throw ConfigError("signal '" + name + "': missing \"" + k + "\"");
    if (c.isMember("rate") == c.isMember("scale")) throw ConfigError("signal '" + name + "': exactly one of \"rate\" and \"scale\"");
    Signal s;
    s.name = name;
    s.title = c.get("title", name.c_str());
    s.filename = c["filename"].asString("filename");
    s.dataset = (unsigned)c["dataset"].asInt("dataset");
    // (-) tells Signal to scale by the total number of samples (config.cpp:216-222)
    double nexpected = c.isMember("rate") ? (double)c["rate"].asFloat("rate") : (double)(-1.0f / c["scale"].asFloat("scale"));
    for (const json::Value& sv : c["systematics"].items) s.systematic_names.push_back(sv.asString("systematics[]"));
    const std::string source_name = c.get("source", name.c_str());
    for (const Source& src : fc.sources)
      if (src.name == source_name) {
        s.source = src;
        break;
      }
    std::vector<float> table;
    if (load_tables) {
      std::vector<float> raw;
      std::vector<std::string> fields, npy_fields;
      for (const json::Value& fv : c["fields"].items) npy_fields.push_back(fv.asString("fields[]"));
      read_table(detail::join_path(base_dir, s.filename), raw, fields, npy_fields);
      s.n_mc = fields.empty() ? 0 : raw.size() / fields.size();   // BEFORE cuts (signal.cpp:28)
      table = read_dataset_to_samples(raw, fields, s.dataset, fc.sample_fields, fc.cuts);
    }
    if (nexpected < 0) nexpected *= -1.0 * (double)s.n_mc;          // signal.cpp:31-35
    s.nexpected = nexpected;
    fc.signals.push_back(s);
    fc.tables.push_back(std::move(table));
  }

  // data sets, clipped to the PDF boundaries: observables act as cuts (config.cpp:260-296)
  const json::Value& data_params = root["data"];
  for (const auto& kv : data_params.members) {
    const unsigned dataset = (unsigned)std::strtoul(kv.first.c_str(), nullptr, 10);
    std::vector<Observable> cc = fc.observables;
    cc.insert(cc.end(), fc.cuts.begin(), fc.cuts.end());
    for (const json::Value& row : kv.second.items) {
      if (!load_tables) continue;
      std::vector<float> raw;
      std::vector<std::string> fields, npy_fields;
      for (const json::Value& fv : row["fields"].items) npy_fields.push_back(fv.asString("fields[]"));
      read_table(detail::join_path(base_dir, row["filename"].asString("filename")), raw, fields, npy_fields);
      const std::vector<float> s = read_dataset_to_samples(raw, fields, dataset, fc.sample_fields, cc, fc.observables.size());
      // GetSamples (pdfz.h:542-556): the observables, then the dataset id
      const size_t D = fc.observables.size();
      fc.data[dataset].emplace_back();
      std::vector<float>& out = fc.data[dataset].back();
      for (size_t r = 0; r < s.size() / fc.nfields; r++) {
        for (size_t k = 0; k < D; k++) out.push_back(s[r * fc.nfields + k]);
        out.push_back(s[r * fc.nfields + fc.nfields - 1]);
      }
    }
  }
  return fc;
}

/** The data experiment i fits when the configuration lists data sets (sxmc.cpp:71-80): file i of EVERY data set, in
 *  the order of the data set ids, appended (GetSamples).  false: no data sets configured -- the caller samples a fake
 *  one.  A data set with fewer than i + 1 files is an error (the reference indexes past the end of its list). */
inline bool experiment_data(const FitConfig& fc, unsigned i, std::vector<float>& rows) {
  if (fc.data.empty()) return false;
  rows.clear();
  for (const auto& kv : fc.data) {
    if (i >= kv.second.size()) {
      throw ConfigError("data set " + std::to_string(kv.first) + " lists " + std::to_string(kv.second.size()) +
                        " file(s): experiment " + std::to_string(i) + " has none to fit (one file per experiment)");
    }
    rows.insert(rows.end(), kv.second[i].begin(), kv.second[i].end());
  }
  return true;
}

/** FitConfig::FitConfig(filename). */
inline FitConfig load_config(const std::string& path, bool load_tables = true) {
  const size_t slash = path.find_last_of('/');
  return parse_config(detail::read_file(path), slash == std::string::npos ? std::string(".") : path.substr(0, slash),
                      load_tables);
}

/** The systematics signal j carries (config.cpp:234-246): those of the union that it lists, in the union's order of
 *  its own list. */
inline std::vector<Systematic> systematics_of(const FitConfig& fc, size_t j) {
  std::vector<Systematic> out;
  for (const std::string& n : fc.signals.at(j).systematic_names)
    for (const Systematic& s : fc.systematics)
      if (s.name == n) out.push_back(s);
  return out;
}

/** true when every signal lists every systematic of the fit in the union's order: what the batched drivers
 *  (one launch for all signals; ensemble_lockstep, ensemble_multi_gpu) assume. */
inline bool same_systematics_everywhere(const FitConfig& fc) {
  for (size_t j = 0; j < fc.signals.size(); j++) {
    const std::vector<Systematic> mine = systematics_of(fc, j);
    if (mine.size() != fc.systematics.size()) return false;
    for (size_t k = 0; k < mine.size(); k++)
      if (mine[k].name != fc.systematics[k].name) return false;
  }
  return true;
}

}  // namespace sxmc
