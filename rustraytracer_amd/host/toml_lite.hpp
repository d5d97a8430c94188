// The subset of TOML the renderer's scene and render configs use (src/main.rs:42-46 parses them with
// the `toml` crate): comments, [tables], [[arrays of tables]] with sub-tables, bare / quoted / dotted
// keys, basic and literal strings, integers, floats, booleans, arrays and inline tables.  No dates,
// no multi-line strings.  Errors carry the line number.
#pragma once
#include <cctype>
#include <cmath>
#include <cstdlib>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace toml_lite {

struct Value {
    enum Kind { NONE, STRING, INT, FLOAT, BOOL, ARRAY, TABLE } kind = NONE;
    std::string s;
    long long i = 0;
    double f = 0.0;
    bool b = false;
    std::vector<Value> a;
    std::map<std::string, Value> t;
    bool inline_table = false;  // closed for later extension, like an inline table in TOML

    const Value* find(const std::string& key) const {
        if (kind != TABLE) return nullptr;
        auto it = t.find(key);
        return it == t.end() ? nullptr : &it->second;
    }
    bool is_number() const { return kind == INT || kind == FLOAT; }
    double number() const { return kind == INT ? (double)i : f; }  // serde's f64 accepts an integer
};

class Parser {
public:
    explicit Parser(const std::string& text) : s_(text) {}

    Value parse() {
        Value root;
        root.kind = Value::TABLE;
        Value* cur = &root;
        for (;;) {
            skip_ws_comments_newlines();
            if (eof()) break;
            if (peek() == '[') {
                ++p_;
                bool array = false;
                if (!eof() && peek() == '[') {
                    array = true;
                    ++p_;
                }
                std::vector<std::string> path = key_path();
                expect(']');
                if (array) expect(']');
                end_of_line();
                cur = open_table(root, path, array);
            } else {
                std::vector<std::string> path = key_path();
                skip_ws();
                expect('=');
                skip_ws();
                Value v = value();
                end_of_line();
                assign(*cur, path, v);
            }
        }
        return root;
    }

private:
    const std::string& s_;
    size_t p_ = 0;
    int line_ = 1;

    [[noreturn]] void fail(const std::string& what) const {
        throw std::runtime_error("TOML parse error at line " + std::to_string(line_) + ": " + what);
    }
    bool eof() const { return p_ >= s_.size(); }
    char peek() const { return s_[p_]; }
    void expect(char c) {
        skip_ws();
        if (eof() || peek() != c) fail(std::string("expected '") + c + "'");
        ++p_;
    }
    void skip_ws() {
        while (!eof() && (peek() == ' ' || peek() == '\t')) ++p_;
    }
    void skip_comment() {
        if (!eof() && peek() == '#')
            while (!eof() && peek() != '\n') ++p_;
    }
    void skip_ws_comments_newlines() {
        for (;;) {
            skip_ws();
            skip_comment();
            if (!eof() && (peek() == '\n' || peek() == '\r')) {
                if (peek() == '\n') ++line_;
                ++p_;
            } else {
                break;
            }
        }
    }
    void end_of_line() {
        skip_ws();
        skip_comment();
        if (eof()) return;
        if (peek() == '\r') ++p_;
        if (eof()) return;
        if (peek() != '\n') fail("expected end of line");
        ++p_;
        ++line_;
    }

    std::string basic_string() {
        std::string out;
        ++p_;  // opening quote
        if (p_ + 1 < s_.size() && s_[p_] == '"' && s_[p_ + 1] == '"') fail("multi-line strings are not supported");
        for (;;) {
            if (eof() || peek() == '\n') fail("unterminated string");
            char c = s_[p_++];
            if (c == '"') break;
            if (c == '\\') {
                if (eof()) fail("unterminated escape");
                char e = s_[p_++];
                switch (e) {
                case 'n': out += '\n'; break;
                case 't': out += '\t'; break;
                case 'r': out += '\r'; break;
                case '"': out += '"'; break;
                case '\\': out += '\\'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                default: fail("unsupported escape sequence");
                }
            } else {
                out += c;
            }
        }
        return out;
    }
    std::string literal_string() {
        std::string out;
        ++p_;
        for (;;) {
            if (eof() || peek() == '\n') fail("unterminated string");
            char c = s_[p_++];
            if (c == '\'') break;
            out += c;
        }
        return out;
    }

    std::string key_part() {
        skip_ws();
        if (eof()) fail("expected a key");
        if (peek() == '"') return basic_string();
        if (peek() == '\'') return literal_string();
        size_t b = p_;
        while (!eof() && (std::isalnum((unsigned char)peek()) || peek() == '_' || peek() == '-')) ++p_;
        if (p_ == b) fail("expected a key");
        return s_.substr(b, p_ - b);
    }
    std::vector<std::string> key_path() {
        std::vector<std::string> path;
        path.push_back(key_part());
        for (;;) {
            skip_ws();
            if (!eof() && peek() == '.') {
                ++p_;
                path.push_back(key_part());
            } else {
                break;
            }
        }
        return path;
    }

    Value value() {
        skip_ws();
        if (eof()) fail("expected a value");
        Value v;
        char c = peek();
        if (c == '"') {
            v.kind = Value::STRING;
            v.s = basic_string();
        } else if (c == '\'') {
            v.kind = Value::STRING;
            v.s = literal_string();
        } else if (c == '[') {
            ++p_;
            v.kind = Value::ARRAY;
            for (;;) {
                skip_ws_comments_newlines();
                if (eof()) fail("unterminated array");
                if (peek() == ']') {
                    ++p_;
                    break;
                }
                v.a.push_back(value());
                skip_ws_comments_newlines();
                if (!eof() && peek() == ',') {
                    ++p_;
                } else {
                    skip_ws_comments_newlines();
                    if (eof() || peek() != ']') fail("expected ',' or ']' in array");
                }
            }
        } else if (c == '{') {
            ++p_;
            v.kind = Value::TABLE;
            v.inline_table = true;
            skip_ws();
            if (!eof() && peek() == '}') {
                ++p_;
                return v;
            }
            for (;;) {
                std::vector<std::string> path = key_path();
                skip_ws();
                expect('=');
                Value item = value();
                assign(v, path, item);
                skip_ws();
                if (eof()) fail("unterminated inline table");
                if (peek() == ',') {
                    ++p_;
                } else if (peek() == '}') {
                    ++p_;
                    break;
                } else {
                    fail("expected ',' or '}' in inline table");
                }
            }
        } else {
            size_t b = p_;
            while (!eof() && !std::isspace((unsigned char)peek()) && peek() != ',' && peek() != ']' && peek() != '}' &&
                   peek() != '#')
                ++p_;
            std::string tok = s_.substr(b, p_ - b);
            if (tok == "true" || tok == "false") {
                v.kind = Value::BOOL;
                v.b = tok == "true";
            } else {
                parse_number(tok, v);
            }
        }
        return v;
    }

    void parse_number(const std::string& tok_in, Value& v) {
        if (tok_in.empty()) fail("expected a value");
        std::string tok;
        for (char c : tok_in)
            if (c != '_') tok += c;
        std::string body = tok;
        double sign = 1.0;
        if (body[0] == '+' || body[0] == '-') {
            if (body[0] == '-') sign = -1.0;
            body = body.substr(1);
        }
        if (body == "inf" || body == "nan") {
            v.kind = Value::FLOAT;
            v.f = body == "inf" ? sign * INFINITY : NAN;
            return;
        }
        if (body.empty() || !std::isdigit((unsigned char)body[0])) fail("unsupported value '" + tok_in + "'");
        bool is_float = false;
        for (char c : body) {
            if (c == '.' || c == 'e' || c == 'E') is_float = true;
            else if (!std::isdigit((unsigned char)c) && c != '+' && c != '-' && c != 'x' && c != 'o' && c != 'b' &&
                     !std::isxdigit((unsigned char)c))
                fail("unsupported value '" + tok_in + "' (dates are not supported)");
        }
        char* end = nullptr;
        std::string digits = tok;  // `end` points into tok or digits: both outlive the check below
        if (is_float && body.compare(0, 2, "0x") != 0) {
            v.kind = Value::FLOAT;
            v.f = std::strtod(tok.c_str(), &end);
        } else {
            v.kind = Value::INT;
            int base = 10;
            if (body.compare(0, 2, "0x") == 0) { base = 16; digits = body.substr(2); }
            else if (body.compare(0, 2, "0o") == 0) { base = 8; digits = body.substr(2); }
            else if (body.compare(0, 2, "0b") == 0) { base = 2; digits = body.substr(2); }
            v.i = std::strtoll(digits.c_str(), &end, base);
        }
        if (!end || *end != '\0') fail("malformed number '" + tok_in + "'");
    }

    // key path inside `table`; intermediate tables are created; the last array-of-tables element is
    // entered when a path component names an array of tables
    Value* descend(Value& table, const std::string& key, bool create) {
        auto it = table.t.find(key);
        if (it == table.t.end()) {
            if (!create) return nullptr;
            Value nv;
            nv.kind = Value::TABLE;
            it = table.t.emplace(key, nv).first;
        }
        Value* v = &it->second;
        if (v->kind == Value::ARRAY) {
            if (v->a.empty() || v->a.back().kind != Value::TABLE) fail("key '" + key + "' is not a table");
            v = &v->a.back();
        }
        if (v->kind != Value::TABLE) fail("key '" + key + "' is not a table");
        if (v->inline_table) fail("inline table '" + key + "' cannot be extended");
        return v;
    }

    Value* open_table(Value& root, const std::vector<std::string>& path, bool array) {
        Value* cur = &root;
        for (size_t k = 0; k + 1 < path.size(); ++k) cur = descend(*cur, path[k], true);
        const std::string& last = path.back();
        if (array) {
            auto it = cur->t.find(last);
            if (it == cur->t.end()) {
                Value arr;
                arr.kind = Value::ARRAY;
                it = cur->t.emplace(last, arr).first;
            }
            if (it->second.kind != Value::ARRAY) fail("'" + last + "' is not an array of tables");
            Value nt;
            nt.kind = Value::TABLE;
            it->second.a.push_back(nt);
            return &it->second.a.back();
        }
        return descend(*cur, last, true);
    }

    void assign(Value& table, const std::vector<std::string>& path, const Value& v) {
        Value* cur = &table;
        bool was_inline = cur->inline_table;
        cur->inline_table = false;  // an inline table is open while its own body is parsed
        for (size_t k = 0; k + 1 < path.size(); ++k) cur = descend(*cur, path[k], true);
        if (cur->t.count(path.back())) fail("duplicate key '" + path.back() + "'");
        cur->t[path.back()] = v;
        table.inline_table = was_inline;
    }
};

inline Value parse(const std::string& text) { return Parser(text).parse(); }

}  // namespace toml_lite
