// prv_json.hpp -- minimal JSON value, parser and styled writer for transforms.json.
// Own implementation (the reference uses JsonCpp, which is not available here); the
// writer follows the layout JsonCpp's StyledWriter produces for the reference's files
// (main.cpp:1648-1650): members sorted by key, 3-space indentation, doubles as %.17g.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace prvjson {

struct Value {
  enum Kind { Null, Bool, Int, Real, String, Array, Object } kind = Null;
  bool b = false;
  long long i = 0;
  double d = 0.0;
  std::string s;
  std::vector<Value> arr;
  std::map<std::string, Value> obj; // sorted, like JsonCpp's ObjectValues

  Value() {}
  Value(bool v) : kind(Bool), b(v) {}
  Value(int v) : kind(Int), i(v) {}
  Value(long long v) : kind(Int), i(v) {}
  Value(double v) : kind(Real), d(v) {}
  Value(const char* v) : kind(String), s(v) {}
  Value(const std::string& v) : kind(String), s(v) {}

  Value& operator[](const std::string& k) {
    if (kind == Null) kind = Object;
    return obj[k];
  }
  Value& operator[](int idx) { // JsonCpp: root["offset"][0] = ... grows the array
    if (kind == Null) kind = Array;
    if ((int)arr.size() <= idx) arr.resize(idx + 1);
    return arr[idx];
  }
  void append(const Value& v) {
    if (kind == Null) kind = Array;
    arr.push_back(v);
  }
  bool has(const std::string& k) const { return kind == Object && obj.count(k); }
  const Value& at(const std::string& k) const {
    static const Value null_value;
    auto it = obj.find(k);
    return it == obj.end() ? null_value : it->second;
  }
  bool is_number() const { return kind == Int || kind == Real; }
  double number() const { return kind == Int ? (double)i : d; }
};

class Parser {
public:
  explicit Parser(const std::string& text) : t_(text) {}
  bool parse(Value& out, std::string& err) {
    try {
      ws();
      out = value();
      ws();
      if (p_ != t_.size()) throw std::string("trailing characters");
      return true;
    } catch (const std::string& e) {
      err = e + " at byte " + std::to_string(p_);
      return false;
    }
  }

private:
  const std::string& t_;
  size_t p_ = 0;
  void ws() {
    while (p_ < t_.size() && (t_[p_] == ' ' || t_[p_] == '\n' || t_[p_] == '\r' || t_[p_] == '\t')) p_++;
  }
  char peek() {
    if (p_ >= t_.size()) throw std::string("unexpected end");
    return t_[p_];
  }
  void expect(char c) {
    if (peek() != c) throw std::string("expected '") + c + "'";
    p_++;
  }
  Value value() {
    char c = peek();
    if (c == '{') return object();
    if (c == '[') return array();
    if (c == '"') return Value(string());
    if (!t_.compare(p_, 4, "true")) { p_ += 4; return Value(true); }
    if (!t_.compare(p_, 5, "false")) { p_ += 5; return Value(false); }
    if (!t_.compare(p_, 4, "null")) { p_ += 4; return Value(); }
    return number();
  }
  Value number() {
    const char* s = t_.c_str() + p_;
    char* e = nullptr;
    double d = strtod(s, &e);
    if (e == s) throw std::string("bad number");
    bool integral = true;
    for (const char* q = s; q < e; q++)
      if (*q == '.' || *q == 'e' || *q == 'E' || *q == 'n' || *q == 'i') integral = false;
    p_ += (size_t)(e - s);
    if (integral && std::fabs(d) < 9e15) return Value((long long)d);
    return Value(d);
  }
  std::string string() {
    expect('"');
    std::string r;
    while (true) {
      char c = peek();
      p_++;
      if (c == '"') break;
      if (c == '\\') {
        char n = peek();
        p_++;
        switch (n) {
          case 'n': r += '\n'; break;
          case 't': r += '\t'; break;
          case 'r': r += '\r'; break;
          case 'b': r += '\b'; break;
          case 'f': r += '\f'; break;
          case 'u': { // keep BMP code points as UTF-8
            if (p_ + 4 > t_.size()) throw std::string("bad \\u escape");
            unsigned cp = (unsigned)strtoul(t_.substr(p_, 4).c_str(), nullptr, 16);
            p_ += 4;
            if (cp < 0x80) r += (char)cp;
            else if (cp < 0x800) { r += (char)(0xC0 | (cp >> 6)); r += (char)(0x80 | (cp & 0x3F)); }
            else { r += (char)(0xE0 | (cp >> 12)); r += (char)(0x80 | ((cp >> 6) & 0x3F)); r += (char)(0x80 | (cp & 0x3F)); }
            break;
          }
          default: r += n;
        }
      } else {
        r += c;
      }
    }
    return r;
  }
  Value array() {
    expect('[');
    Value v;
    v.kind = Value::Array;
    ws();
    if (peek() == ']') { p_++; return v; }
    while (true) {
      ws();
      v.arr.push_back(value());
      ws();
      if (peek() == ',') { p_++; continue; }
      expect(']');
      break;
    }
    return v;
  }
  Value object() {
    expect('{');
    Value v;
    v.kind = Value::Object;
    ws();
    if (peek() == '}') { p_++; return v; }
    while (true) {
      ws();
      std::string k = string();
      ws();
      expect(':');
      ws();
      v.obj[k] = value();
      ws();
      if (peek() == ',') { p_++; continue; }
      expect('}');
      break;
    }
    return v;
  }
};

inline std::string real_to_string(double d) {
  char buf[64];
  if (std::isnan(d)) return "null";
  if (std::isinf(d)) return d < 0 ? "-1e+9999" : "1e+9999";
  snprintf(buf, sizeof(buf), "%.17g", d);
  if (!strchr(buf, '.') && !strchr(buf, 'e')) strcat(buf, ".0");
  return buf;
}

inline std::string quote(const std::string& s) {
  std::string r = "\"";
  for (char c : s) {
    switch (c) {
      case '"': r += "\\\""; break;
      case '\\': r += "\\\\"; break;
      case '\n': r += "\\n"; break;
      case '\t': r += "\\t"; break;
      case '\r': r += "\\r"; break;
      default: r += c;
    }
  }
  return r + "\"";
}

inline std::string scalar_to_string(const Value& v) {
  switch (v.kind) {
    case Value::Null: return "null";
    case Value::Bool: return v.b ? "true" : "false";
    case Value::Int: return std::to_string(v.i);
    case Value::Real: return real_to_string(v.d);
    case Value::String: return quote(v.s);
    default: return "";
  }
}

// StyledWriter-like: arrays go on one line ("[ a, b, c ]") when all children are scalars
// and the line stays within 74 columns, otherwise one element per line.
inline void write_styled(const Value& v, std::string& out, int indent = 0) {
  const std::string pad((size_t)indent, ' ');
  if (v.kind == Value::Object) {
    if (v.obj.empty()) { out += "{}"; return; }
    out += "{\n";
    size_t n = 0;
    for (const auto& kv : v.obj) {
      out += pad + "   " + quote(kv.first) + " : ";
      write_styled(kv.second, out, indent + 3);
      if (++n != v.obj.size()) out += ",";
      out += "\n";
    }
    out += pad + "}";
  } else if (v.kind == Value::Array) {
    if (v.arr.empty()) { out += "[]"; return; }
    bool multi = false;
    size_t len = 4 + (v.arr.size() - 1) * 2;
    for (const auto& c : v.arr) {
      if ((c.kind == Value::Array || c.kind == Value::Object) && !(c.arr.empty() && c.obj.empty())) multi = true;
      else len += scalar_to_string(c).size();
    }
    if (len >= 74) multi = true;
    if (!multi) {
      out += "[ ";
      for (size_t k = 0; k < v.arr.size(); k++) {
        if (k) out += ", ";
        out += scalar_to_string(v.arr[k]);
      }
      out += " ]";
    } else {
      out += "[\n";
      for (size_t k = 0; k < v.arr.size(); k++) {
        out += pad + "   ";
        write_styled(v.arr[k], out, indent + 3);
        if (k + 1 != v.arr.size()) out += ",";
        out += "\n";
      }
      out += pad + "]";
    }
  } else {
    out += scalar_to_string(v);
  }
}

inline std::string to_styled_string(const Value& v) {
  std::string s;
  write_styled(v, s, 0);
  s += "\n";
  return s;
}

inline bool read_file(const std::string& path, std::string& text) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  char buf[65536];
  size_t n;
  text.clear();
  while ((n = fread(buf, 1, sizeof(buf), f)) > 0) text.append(buf, n);
  fclose(f);
  return true;
}

} // namespace prvjson
