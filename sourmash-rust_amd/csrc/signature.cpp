// signature.cpp -- the Signature container and its JSON wire format (".sig" v0.4).
//
// Host only.  Restates reference src/lib.rs:62-139 (KmerMinHash Serialize / Deserialize incl.
// the md5sum of ksize + decimal mins), 546-675 (Signature, load_signatures, PartialEq).  The
// writer reproduces serde_json's compact output (field order of the Serialize impls, no spaces).
#include "signature.hpp"

#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

namespace smh {

// ------------------------------------------------------------------------------------
// MD5 (RFC 1321) -- only for the md5sum field of serialised sketches (src/lib.rs:72-77,86)
namespace {

struct Md5 {
  uint32_t a = 0x67452301, b = 0xefcdab89, c = 0x98badcfe, d = 0x10325476;
  uint64_t total = 0;
  uint8_t buf[64];
  size_t fill = 0;

  static uint32_t rol(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }

  void block(const uint8_t* p) {
    static const uint32_t K[64] = {
        0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501,
        0x698098d8, 0x8b44f7af, 0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821,
        0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8,
        0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a,
        0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70,
        0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665,
        0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1,
        0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
    static const int S[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22,
                              5, 9,  14, 20, 5, 9,  14, 20, 5, 9,  14, 20, 5, 9,  14, 20,
                              4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23,
                              6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
    uint32_t M[16];
    for (int i = 0; i < 16; i++)
      M[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) |
             ((uint32_t)p[4 * i + 3] << 24);
    uint32_t A = a, B = b, C = c, D = d;
    for (int i = 0; i < 64; i++) {
      uint32_t F;
      int g;
      if (i < 16) { F = (B & C) | (~B & D); g = i; }
      else if (i < 32) { F = (D & B) | (~D & C); g = (5 * i + 1) & 15; }
      else if (i < 48) { F = B ^ C ^ D; g = (3 * i + 5) & 15; }
      else { F = C ^ (B | ~D); g = (7 * i) & 15; }
      F = F + A + K[i] + M[g];
      A = D; D = C; C = B;
      B = B + rol(F, S[i]);
    }
    a += A; b += B; c += C; d += D;
  }

  void update(const void* data, size_t n) {
    const uint8_t* p = (const uint8_t*)data;
    total += n;
    while (n) {
      size_t take = std::min(n, (size_t)64 - fill);
      memcpy(buf + fill, p, take);
      fill += take; p += take; n -= take;
      if (fill == 64) { block(buf); fill = 0; }
    }
  }

  std::string hex() {
    uint64_t bits = total * 8;
    uint8_t pad = 0x80;
    update(&pad, 1);
    uint8_t z = 0;
    while (fill != 56) update(&z, 1);
    uint8_t len[8];
    for (int i = 0; i < 8; i++) len[i] = (uint8_t)(bits >> (8 * i));
    update(len, 8);
    uint32_t w[4] = {a, b, c, d};
    char out[33];
    for (int i = 0; i < 16; i++) snprintf(out + 2 * i, 3, "%02x", (w[i / 4] >> (8 * (i % 4))) & 0xff);
    return std::string(out, 32);
  }
};

}  // namespace

std::string sketch_md5(const KmerMinHash& mh) {
  mh.materialize();
  Md5 m;
  std::string k = std::to_string(mh.ksize);
  m.update(k.data(), k.size());
  char tmp[24];
  for (uint64_t v : mh.mins) {
    int n = snprintf(tmp, sizeof tmp, "%llu", (unsigned long long)v);
    m.update(tmp, (size_t)n);
  }
  return m.hex();
}

// ------------------------------------------------------------------------------------
// minimal JSON document model + parser (what serde_json accepts for these types)
namespace {

struct JVal {
  enum Kind { Null, Bool, UInt, NegInt, Float, Str, Arr, Obj } kind = Null;
  bool b = false;
  uint64_t u = 0;     // UInt: value; NegInt: magnitude
  double f = 0.0;
  std::string s;
  std::vector<JVal> arr;
  std::vector<std::pair<std::string, JVal>> obj;
  const JVal* get(const char* key) const {
    const JVal* found = nullptr;
    for (auto& kv : obj) if (kv.first == key) found = &kv.second;  // serde: duplicate -> error; keep last
    return found;
  }
};

[[noreturn]] void serde_error(const std::string& m) { throw Error(kSerdeError, m); }

struct JParser {
  const char* p;
  const char* end;
  int depth = 0;
  void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++; }
  [[noreturn]] void fail(const char* what) { serde_error(std::string("JSON error: ") + what); }

  void parse_string(std::string& out) {
    if (p >= end || *p != '"') fail("expected string");
    p++;
    while (true) {
      if (p >= end) fail("EOF while parsing a string");
      unsigned char c = (unsigned char)*p++;
      if (c == '"') break;
      if (c < 0x20) fail("control character in string");
      if (c != '\\') { out.push_back((char)c); continue; }
      if (p >= end) fail("EOF in escape");
      char e = *p++;
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
          auto hex4 = [&]() {
            if (end - p < 4) fail("EOF in \\u escape");
            uint32_t v = 0;
            for (int i = 0; i < 4; i++) {
              char h = *p++;
              v <<= 4;
              if (h >= '0' && h <= '9') v |= h - '0';
              else if (h >= 'a' && h <= 'f') v |= h - 'a' + 10;
              else if (h >= 'A' && h <= 'F') v |= h - 'A' + 10;
              else fail("invalid escape");
            }
            return v;
          };
          uint32_t cp = hex4();
          if (cp >= 0xD800 && cp <= 0xDBFF) {
            if (end - p < 2 || p[0] != '\\' || p[1] != 'u') fail("lone surrogate");
            p += 2;
            uint32_t lo = hex4();
            if (lo < 0xDC00 || lo > 0xDFFF) fail("lone surrogate");
            cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
          } else if (cp >= 0xDC00 && cp <= 0xDFFF) fail("lone surrogate");
          if (cp < 0x80) out.push_back((char)cp);
          else if (cp < 0x800) { out.push_back((char)(0xC0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3F))); }
          else if (cp < 0x10000) {
            out.push_back((char)(0xE0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
            out.push_back((char)(0x80 | (cp & 0x3F)));
          } else {
            out.push_back((char)(0xF0 | (cp >> 18))); out.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
            out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F)));
          }
          break;
        }
        default: fail("invalid escape");
      }
    }
  }

  void parse_number(JVal& v) {
    const char* s = p;
    bool neg = false;
    if (*p == '-') { neg = true; p++; }
    if (p >= end || !isdigit((unsigned char)*p)) fail("invalid number");
    bool is_float = false, overflow = false;
    uint64_t acc = 0;
    if (*p == '0') { p++; }
    else while (p < end && isdigit((unsigned char)*p)) {
      uint64_t d = (uint64_t)(*p - '0');
      if (acc > (UINT64_MAX - d) / 10) overflow = true; else acc = acc * 10 + d;
      p++;
    }
    if (p < end && *p == '.') { is_float = true; p++; if (p >= end || !isdigit((unsigned char)*p)) fail("invalid number"); while (p < end && isdigit((unsigned char)*p)) p++; }
    if (p < end && (*p == 'e' || *p == 'E')) {
      is_float = true; p++;
      if (p < end && (*p == '+' || *p == '-')) p++;
      if (p >= end || !isdigit((unsigned char)*p)) fail("invalid number");
      while (p < end && isdigit((unsigned char)*p)) p++;
    }
    if (is_float || overflow) { v.kind = JVal::Float; v.f = strtod(std::string(s, p).c_str(), nullptr); }
    else if (neg) { v.kind = JVal::NegInt; v.u = acc; v.f = -(double)acc; }
    else { v.kind = JVal::UInt; v.u = acc; v.f = (double)acc; }
  }

  void parse_value(JVal& v) {
    ws();
    if (p >= end) fail("EOF while parsing a value");
    if (++depth > 128) fail("recursion limit exceeded");
    char c = *p;
    if (c == '{') {
      v.kind = JVal::Obj; p++; ws();
      if (p < end && *p == '}') { p++; }
      else while (true) {
        ws();
        std::string k; parse_string(k);
        ws();
        if (p >= end || *p != ':') fail("expected ':'");
        p++;
        v.obj.emplace_back(std::move(k), JVal());
        parse_value(v.obj.back().second);
        ws();
        if (p < end && *p == ',') { p++; continue; }
        if (p < end && *p == '}') { p++; break; }
        fail("expected ',' or '}'");
      }
    } else if (c == '[') {
      v.kind = JVal::Arr; p++; ws();
      if (p < end && *p == ']') { p++; }
      else while (true) {
        v.arr.emplace_back();
        parse_value(v.arr.back());
        ws();
        if (p < end && *p == ',') { p++; continue; }
        if (p < end && *p == ']') { p++; break; }
        fail("expected ',' or ']'");
      }
    } else if (c == '"') { v.kind = JVal::Str; parse_string(v.s); }
    else if (c == 't' && end - p >= 4 && !memcmp(p, "true", 4)) { v.kind = JVal::Bool; v.b = true; p += 4; }
    else if (c == 'f' && end - p >= 5 && !memcmp(p, "false", 5)) { v.kind = JVal::Bool; v.b = false; p += 5; }
    else if (c == 'n' && end - p >= 4 && !memcmp(p, "null", 4)) { v.kind = JVal::Null; p += 4; }
    else if (c == '-' || isdigit((unsigned char)c)) parse_number(v);
    else fail("expected value");
    depth--;
  }
};

uint64_t want_uint(const JVal* v, const char* field, uint64_t maxv) {
  if (!v) serde_error(std::string("missing field `") + field + "`");
  if (v->kind != JVal::UInt || v->u > maxv) serde_error(std::string("invalid type for field `") + field + "`");
  return v->u;
}
const std::string& want_str(const JVal* v, const char* field) {
  if (!v) serde_error(std::string("missing field `") + field + "`");
  if (v->kind != JVal::Str) serde_error(std::string("invalid type for field `") + field + "`: expected a string");
  return v->s;
}
std::vector<uint64_t> want_u64_array(const JVal* v, const char* field) {
  if (!v) serde_error(std::string("missing field `") + field + "`");
  if (v->kind != JVal::Arr) serde_error(std::string("invalid type for field `") + field + "`: expected a sequence");
  std::vector<uint64_t> out;
  out.reserve(v->arr.size());
  for (auto& e : v->arr) {
    if (e.kind != JVal::UInt) serde_error(std::string("invalid type in `") + field + "`: expected u64");
    out.push_back(e.u);
  }
  return out;
}

// reference src/lib.rs:104-139
KmerMinHash sketch_from_json(const JVal& v) {
  if (v.kind != JVal::Obj) serde_error("invalid type: expected struct TempSig");
  KmerMinHash mh;
  uint32_t num = (uint32_t)want_uint(v.get("num"), "num", UINT32_MAX);
  mh.ksize = (uint32_t)want_uint(v.get("ksize"), "ksize", UINT32_MAX);
  mh.seed = want_uint(v.get("seed"), "seed", UINT64_MAX);
  mh.max_hash = want_uint(v.get("max_hash"), "max_hash", UINT64_MAX);
  (void)want_str(v.get("md5sum"), "md5sum");
  mh.mins = want_u64_array(v.get("mins"), "mins");
  const JVal* ab = v.get("abundances");
  if (ab && ab->kind != JVal::Null) { mh.has_abunds = true; mh.abunds = want_u64_array(ab, "abundances"); }
  else { mh.has_abunds = false; mh.abunds.clear(); }
  const std::string& mol = want_str(v.get("molecule"), "molecule");
  mh.is_protein = mol == "protein";            // anything else reads as DNA (Q9)
  mh.num = mh.max_hash != 0 ? 0 : num;          // Q9
  return mh;
}

// reference src/lib.rs:546-577
Signature signature_from_json(const JVal& v) {
  if (v.kind != JVal::Obj) serde_error("invalid type: expected struct Signature");
  Signature s;
  if (const JVal* c = v.get("class")) s.klass = want_str(c, "class");
  if (const JVal* e = v.get("email")) s.email = want_str(e, "email");
  s.hash_function = want_str(v.get("hash_function"), "hash_function");
  if (const JVal* f = v.get("filename")) { if (f->kind != JVal::Null) { s.has_filename = true; s.filename = want_str(f, "filename"); } }
  if (const JVal* n = v.get("name")) { if (n->kind != JVal::Null) { s.has_name = true; s.name = want_str(n, "name"); } }
  if (const JVal* l = v.get("license")) s.license = want_str(l, "license");
  const JVal* sk = v.get("signatures");
  if (!sk) serde_error("missing field `signatures`");
  if (sk->kind != JVal::Arr) serde_error("invalid type for field `signatures`: expected a sequence");
  for (auto& e : sk->arr) s.signatures.push_back(sketch_from_json(e));
  if (const JVal* ver = v.get("version")) {
    if (ver->kind == JVal::UInt || ver->kind == JVal::NegInt || ver->kind == JVal::Float) s.version = ver->f;
    else serde_error("invalid type for field `version`: expected f64");
  }
  return s;
}

void json_escape(std::string& out, const std::string& s) {
  out.push_back('"');
  for (unsigned char c : s) {
    switch (c) {
      case '"': out += "\\\""; break;
      case '\\': out += "\\\\"; break;
      case '\b': out += "\\b"; break;
      case '\f': out += "\\f"; break;
      case '\n': out += "\\n"; break;
      case '\r': out += "\\r"; break;
      case '\t': out += "\\t"; break;
      default:
        if (c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); out += b; }
        else out.push_back((char)c);
    }
  }
  out.push_back('"');
}

// shortest decimal that round-trips, in serde_json's (ryu) style for the values a version takes
std::string f64_to_json(double v) {
  if (!std::isfinite(v)) return "null";
  char b[40];
  for (int prec = 1; prec <= 17; prec++) {
    snprintf(b, sizeof b, "%.*g", prec, v);
    if (strtod(b, nullptr) == v) break;
  }
  std::string s(b);
  if (s.find_first_of(".eEn") == std::string::npos) s += ".0";
  return s;
}

void u64_array(std::string& out, const std::vector<uint64_t>& v) {
  out.push_back('[');
  char tmp[24];
  for (size_t i = 0; i < v.size(); i++) {
    if (i) out.push_back(',');
    int n = snprintf(tmp, sizeof tmp, "%llu", (unsigned long long)v[i]);
    out.append(tmp, (size_t)n);
  }
  out.push_back(']');
}

// reference src/lib.rs:62-102
void sketch_to_json(std::string& out, const KmerMinHash& mh) {
  mh.materialize();
  out += "{\"num\":" + std::to_string(mh.num);
  out += ",\"ksize\":" + std::to_string(mh.ksize);
  out += ",\"seed\":" + std::to_string(mh.seed);
  out += ",\"max_hash\":" + std::to_string(mh.max_hash);
  out += ",\"mins\":";
  u64_array(out, mh.mins);
  out += ",\"md5sum\":\"" + sketch_md5(mh) + "\"";
  if (mh.has_abunds) { out += ",\"abundances\":"; u64_array(out, mh.abunds); }
  out += std::string(",\"molecule\":\"") + (mh.is_protein ? "protein" : "DNA") + "\"}";
}

}  // namespace

bool sketch_equal(const KmerMinHash& a, const KmerMinHash& b) {
  a.materialize();
  b.materialize();
  return a.num == b.num && a.ksize == b.ksize && a.is_protein == b.is_protein && a.seed == b.seed &&
         a.max_hash == b.max_hash && a.mins.get() == b.mins.get() && a.has_abunds == b.has_abunds &&
         (!a.has_abunds || a.abunds == b.abunds);
}

// derive(Serialize) field order of reference src/lib.rs:546-565
void signature_to_json(std::string& out, const Signature& s) {
  out += "{\"class\":"; json_escape(out, s.klass);
  out += ",\"email\":"; json_escape(out, s.email);
  out += ",\"hash_function\":"; json_escape(out, s.hash_function);
  out += ",\"filename\":"; if (s.has_filename) json_escape(out, s.filename); else out += "null";
  out += ",\"name\":"; if (s.has_name) json_escape(out, s.name); else out += "null";
  out += ",\"license\":"; json_escape(out, s.license);
  out += ",\"signatures\":[";
  for (size_t i = 0; i < s.signatures.size(); i++) {
    if (i) out.push_back(',');
    sketch_to_json(out, s.signatures[i]);
  }
  out += "],\"version\":" + f64_to_json(s.version) + "}";
}

std::string signatures_to_json(const std::vector<const Signature*>& v) {
  std::string out = "[";
  for (size_t i = 0; i < v.size(); i++) {
    if (i) out.push_back(',');
    signature_to_json(out, *v[i]);
  }
  out.push_back(']');
  return out;
}

// reference src/lib.rs:585-591 from_reader: a JSON array of signatures
std::vector<Signature> signatures_from_json(const char* data, size_t len) {
  JParser ps{data, data + len};
  JVal root;
  ps.parse_value(root);
  ps.ws();
  if (ps.p != ps.end) serde_error("JSON error: trailing characters");
  if (root.kind != JVal::Arr) serde_error("invalid type: expected a sequence");
  std::vector<Signature> out;
  out.reserve(root.arr.size());
  for (auto& e : root.arr) out.push_back(signature_from_json(e));
  return out;
}

// reference src/lib.rs:593-645 load_signatures: one Signature per sketch, filtered
std::vector<Signature> load_signatures(const char* data, size_t len, size_t ksize, const char* moltype) {
  std::vector<Signature> orig = signatures_from_json(data, len);
  std::string mt;
  if (moltype) for (const char* c = moltype; *c; c++) mt.push_back((char)tolower((unsigned char)*c));
  std::vector<Signature> out;
  for (auto& s : orig) {
    for (auto& mh : s.signatures) {
      if (!(ksize == 0 || ksize == (size_t)mh.ksize)) continue;
      if (moltype && !((mt == "dna" && !mh.is_protein) || (mt == "protein" && mh.is_protein))) continue;
      Signature one = s;
      one.signatures.clear();
      one.signatures.push_back(mh);
      out.push_back(std::move(one));
    }
  }
  return out;
}

bool signature_equal(const Signature& a, const Signature& b) {
  // reference src/lib.rs:663-675: metadata + first sketch; indexing an empty list panics
  if (a.signatures.empty() || b.signatures.empty()) throw_panic("index out of bounds: the len is 0 but the index is 0");
  bool meta = a.klass == b.klass && a.email == b.email && a.hash_function == b.hash_function &&
              a.has_filename == b.has_filename && (!a.has_filename || a.filename == b.filename) &&
              a.has_name == b.has_name && (!a.has_name || a.name == b.name);
  return meta && sketch_equal(a.signatures[0], b.signatures[0]);
}

std::string read_file(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw Error(kIo, "No such file or directory (os error 2)");
  std::stringstream ss;
  ss << f.rdbuf();
  return ss.str();
}

}  // namespace smh
