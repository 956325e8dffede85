// ron.hpp -- a small reader for the subset of RON the reference's scene files use
// (serde + ron 0.8, schema in src/io.rs:24-94): named and anonymous structs
// `Name(field: value, ...)`, tuples `(a, b)`, enum variants `Variant(a, b)`,
// `Some(x)` / `None`, maps `{ "key": value }`, lists `[a, b]`, strings with escapes,
// numbers, booleans, `//` and `/* */` comments, trailing commas.
#ifndef RT_HOST_RON_HPP
#define RT_HOST_RON_HPP

#include <cctype>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace ron {

struct Value;
using ValuePtr = std::shared_ptr<Value>;

struct Value {
    enum Kind { Number, String, Bool, Ident, Tuple, List, Map } kind = Ident;
    double number = 0.0;
    bool boolean = false;
    std::string text;  // String: content; Ident: the identifier; Tuple: optional leading name
    // Tuple: fields with optional names; List: items (names empty); Map: key in `keys`
    std::vector<std::string> names;
    std::vector<ValuePtr> items;
    std::vector<ValuePtr> keys;

    const Value *field(const std::string &name) const {
        for (size_t i = 0; i < items.size(); i++)
            if (i < names.size() && names[i] == name) return items[i].get();
        return nullptr;
    }
    bool is_none() const { return kind == Ident && text == "None"; }
    // Option<T>: `Some(x)` -> x, `None` -> nullptr; anything else is taken as the value itself
    const Value *unwrap_option() const {
        if (is_none()) return nullptr;
        if (kind == Tuple && text == "Some" && items.size() == 1) return items[0].get();
        return this;
    }
};

class Parser {
  public:
    explicit Parser(const std::string &src) : s_(src) {}
    ValuePtr parse_document() {
        ValuePtr v = parse_value();
        skip();
        if (pos_ != s_.size()) fail("trailing characters");
        return v;
    }

  private:
    const std::string &s_;
    size_t pos_ = 0;

    [[noreturn]] void fail(const std::string &what) const {
        size_t line = 1, col = 1;
        for (size_t i = 0; i < pos_ && i < s_.size(); i++) {
            if (s_[i] == '\n') {
                line++;
                col = 1;
            } else {
                col++;
            }
        }
        throw std::runtime_error("RON parse error at " + std::to_string(line) + ":" + std::to_string(col) + ": " + what);
    }
    void skip() {
        for (;;) {
            while (pos_ < s_.size() && std::isspace((unsigned char)s_[pos_])) pos_++;
            if (pos_ + 1 < s_.size() && s_[pos_] == '/' && s_[pos_ + 1] == '/') {
                while (pos_ < s_.size() && s_[pos_] != '\n') pos_++;
            } else if (pos_ + 1 < s_.size() && s_[pos_] == '/' && s_[pos_ + 1] == '*') {
                size_t e = s_.find("*/", pos_ + 2);
                if (e == std::string::npos) fail("unterminated comment");
                pos_ = e + 2;
            } else {
                return;
            }
        }
    }
    bool eat(char c) {
        skip();
        if (pos_ < s_.size() && s_[pos_] == c) {
            pos_++;
            return true;
        }
        return false;
    }
    char peek() {
        skip();
        return pos_ < s_.size() ? s_[pos_] : '\0';
    }
    std::string parse_string_body() {
        std::string out;
        pos_++;  // opening quote
        while (pos_ < s_.size() && s_[pos_] != '"') {
            char c = s_[pos_++];
            if (c == '\\' && pos_ < s_.size()) {
                char e = s_[pos_++];
                switch (e) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case '0': out += '\0'; break;
                    default: out += e; break;  // \\ \" \'
                }
            } else {
                out += c;
            }
        }
        if (pos_ >= s_.size()) fail("unterminated string");
        pos_++;
        return out;
    }
    // the inside of (...) after the opening parenthesis
    void parse_fields(Value &v) {
        for (;;) {
            if (eat(')')) return;
            // `name: value` or just `value`
            size_t save = pos_;
            skip();
            std::string name;
            if (pos_ < s_.size() && (std::isalpha((unsigned char)s_[pos_]) || s_[pos_] == '_')) {
                size_t p = pos_;
                while (p < s_.size() && (std::isalnum((unsigned char)s_[p]) || s_[p] == '_')) p++;
                size_t q = p;
                while (q < s_.size() && std::isspace((unsigned char)s_[q])) q++;
                if (q < s_.size() && s_[q] == ':') {
                    name = s_.substr(pos_, p - pos_);
                    pos_ = q + 1;
                } else {
                    pos_ = save;
                }
            }
            v.names.push_back(name);
            v.items.push_back(parse_value());
            if (eat(',')) continue;
            if (eat(')')) return;
            fail("expected ',' or ')'");
        }
    }
    ValuePtr parse_value() {
        auto v = std::make_shared<Value>();
        char c = peek();
        if (c == '"') {
            v->kind = Value::String;
            v->text = parse_string_body();
        } else if (c == '(') {
            pos_++;
            v->kind = Value::Tuple;
            parse_fields(*v);
        } else if (c == '[') {
            pos_++;
            v->kind = Value::List;
            for (;;) {
                if (eat(']')) break;
                v->items.push_back(parse_value());
                if (eat(',')) continue;
                if (eat(']')) break;
                fail("expected ',' or ']'");
            }
        } else if (c == '{') {
            pos_++;
            v->kind = Value::Map;
            for (;;) {
                if (eat('}')) break;
                v->keys.push_back(parse_value());
                if (!eat(':')) fail("expected ':' in map");
                v->items.push_back(parse_value());
                if (eat(',')) continue;
                if (eat('}')) break;
                fail("expected ',' or '}'");
            }
        } else if (std::isdigit((unsigned char)c) || c == '-' || c == '+' || c == '.') {
            const char *b = s_.c_str() + pos_;
            char *e = nullptr;
            v->kind = Value::Number;
            v->number = std::strtod(b, &e);
            if (e == b) fail("bad number");
            pos_ += (size_t)(e - b);
        } else if (std::isalpha((unsigned char)c) || c == '_') {
            size_t p = pos_;
            while (p < s_.size() && (std::isalnum((unsigned char)s_[p]) || s_[p] == '_')) p++;
            std::string id = s_.substr(pos_, p - pos_);
            pos_ = p;
            if (id == "true" || id == "false") {
                v->kind = Value::Bool;
                v->boolean = id == "true";
            } else if (id == "inf" || id == "NaN") {
                v->kind = Value::Number;
                v->number = id == "inf" ? HUGE_VAL : std::strtod("nan", nullptr);
            } else if (peek() == '(') {
                pos_++;
                v->kind = Value::Tuple;
                v->text = id;
                parse_fields(*v);
            } else {
                v->kind = Value::Ident;
                v->text = id;
            }
        } else {
            fail(std::string("unexpected character '") + c + "'");
        }
        return v;
    }
};

inline ValuePtr parse(const std::string &src) { return Parser(src).parse_document(); }

}  // namespace ron
#endif
