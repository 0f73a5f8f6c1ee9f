// Minimal JSON reader under the include name the StencilStream examples use.
//
// The reference fetches nlohmann/json v3.11.3 at configure time (examples/fdtd/CMakeLists.txt:9-14);
// this build has no network, so the subset of that API which the FDTD and convection examples
// touch is provided here: json::parse(istream), contains, operator[] / at (objects and arrays),
// is_number / is_object / is_array / is_string, type_name, get<T>(), implicit conversion to
// arithmetic types and std::string, size(), range-for over arrays, and parse_error with what().
// Numbers are parsed with strtod / strtoll, like nlohmann does, so converted values are identical.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <istream>
#include <iterator>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

namespace nlohmann {

namespace detail {
class exception : public std::exception {
  public:
    exception(int id, std::string const &text) : id(id), message(text) {}
    const char *what() const noexcept override { return message.c_str(); }
    const int id;

  private:
    std::string message;
};

class parse_error : public exception {
  public:
    parse_error(std::size_t byte, std::string const &text)
        : exception(101, "[json.exception.parse_error.101] parse error at byte " +
                             std::to_string(byte) + ": " + text),
          byte(byte) {}
    const std::size_t byte;
};

class type_error : public exception {
  public:
    explicit type_error(std::string const &text)
        : exception(302, "[json.exception.type_error.302] " + text) {}
};

class out_of_range : public exception {
  public:
    explicit out_of_range(std::string const &text)
        : exception(403, "[json.exception.out_of_range.403] " + text) {}
};
} // namespace detail

class json {
  public:
    enum class value_t { null, object, array, string, boolean, number_integer, number_float };

    using exception = detail::exception;
    using parse_error = detail::parse_error;
    using type_error = detail::type_error;
    using out_of_range = detail::out_of_range;
    using object_t = std::map<std::string, json>;
    using array_t = std::vector<json>;
    using iterator = array_t::iterator;
    using const_iterator = array_t::const_iterator;

    json() = default;
    json(std::nullptr_t) {}
    // value semantics: copies are deep
    json(json const &o)
        : kind(o.kind), boolean(o.boolean), integer(o.integer), real(o.real), text(o.text),
          members(o.members ? std::make_shared<object_t>(*o.members) : nullptr),
          elements(o.elements ? std::make_shared<array_t>(*o.elements) : nullptr) {}
    json(json &&) = default;
    json &operator=(json o) {
        kind = o.kind;
        boolean = o.boolean;
        integer = o.integer;
        real = o.real;
        text = std::move(o.text);
        members = std::move(o.members);
        elements = std::move(o.elements);
        return *this;
    }
    json(bool b) : kind(value_t::boolean), boolean(b) {}
    json(double d) : kind(value_t::number_float), real(d) {}
    json(std::int64_t i) : kind(value_t::number_integer), integer(i) {}
    json(int i) : json(std::int64_t(i)) {}
    json(std::string s) : kind(value_t::string), text(std::move(s)) {}
    json(const char *s) : json(std::string(s)) {}

    // ---- parsing
    static json parse(std::istream &in) {
        std::string all((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        return parse(all);
    }
    static json parse(std::istream &&in) { return parse(in); }
    static json parse(std::string const &source) {
        Parser p{source, 0};
        json value = p.value();
        p.skip_space();
        if (p.at != source.size())
            throw parse_error(p.at + 1, "unexpected trailing characters");
        return value;
    }

    // ---- inspection
    value_t type() const { return kind; }
    bool is_null() const { return kind == value_t::null; }
    bool is_object() const { return kind == value_t::object; }
    bool is_array() const { return kind == value_t::array; }
    bool is_string() const { return kind == value_t::string; }
    bool is_boolean() const { return kind == value_t::boolean; }
    bool is_number() const {
        return kind == value_t::number_integer || kind == value_t::number_float;
    }
    bool is_number_integer() const { return kind == value_t::number_integer; }
    bool is_number_float() const { return kind == value_t::number_float; }
    const char *type_name() const {
        switch (kind) {
        case value_t::null:
            return "null";
        case value_t::object:
            return "object";
        case value_t::array:
            return "array";
        case value_t::string:
            return "string";
        case value_t::boolean:
            return "boolean";
        default:
            return "number";
        }
    }

    std::size_t size() const {
        if (kind == value_t::object)
            return members ? members->size() : 0;
        if (kind == value_t::array)
            return elements ? elements->size() : 0;
        return kind == value_t::null ? 0 : 1;
    }
    bool empty() const { return size() == 0; }

    bool contains(std::string const &key) const {
        return kind == value_t::object && members && members->count(key) != 0;
    }
    std::size_t count(std::string const &key) const { return contains(key) ? 1 : 0; }

    // ---- element access
    json &operator[](std::string const &key) {
        if (kind == value_t::null) {
            kind = value_t::object;
        }
        if (kind != value_t::object)
            throw type_error(std::string("cannot use operator[] with a string argument with ") +
                             type_name());
        if (!members)
            members = std::make_shared<object_t>();
        return (*members)[key];
    }
    json &operator[](const char *key) { return (*this)[std::string(key)]; }
    json const &operator[](std::string const &key) const { return at(key); }
    json &operator[](std::size_t index) { return at(index); }
    json const &operator[](std::size_t index) const { return at(index); }
    json &operator[](int index) { return at(std::size_t(index)); }

    json &at(std::string const &key) {
        if (kind != value_t::object)
            throw type_error(std::string("cannot use at() with ") + type_name());
        auto it = members ? members->find(key) : object_t::iterator();
        if (!members || it == members->end())
            throw out_of_range("key '" + key + "' not found");
        return it->second;
    }
    json const &at(std::string const &key) const { return const_cast<json *>(this)->at(key); }
    json &at(std::size_t index) {
        if (kind != value_t::array)
            throw type_error(std::string("cannot use at() with ") + type_name());
        if (!elements || index >= elements->size())
            throw out_of_range("array index " + std::to_string(index) + " is out of range");
        return (*elements)[index];
    }
    json const &at(std::size_t index) const { return const_cast<json *>(this)->at(index); }

    iterator begin() { return array_storage().begin(); }
    iterator end() { return array_storage().end(); }
    const_iterator begin() const { return const_cast<json *>(this)->array_storage().begin(); }
    const_iterator end() const { return const_cast<json *>(this)->array_storage().end(); }

    // ---- conversion
    template <typename T> T get() const {
        if constexpr (std::is_same_v<T, std::string>) {
            if (kind != value_t::string)
                throw type_error(std::string("type must be string, but is ") + type_name());
            return text;
        } else if constexpr (std::is_same_v<T, bool>) {
            if (kind != value_t::boolean)
                throw type_error(std::string("type must be boolean, but is ") + type_name());
            return boolean;
        } else {
            static_assert(std::is_arithmetic_v<T>, "unsupported conversion");
            switch (kind) {
            case value_t::number_integer:
                return static_cast<T>(integer);
            case value_t::number_float:
                return static_cast<T>(real);
            case value_t::boolean:
                return static_cast<T>(boolean);
            default:
                throw type_error(std::string("type must be number, but is ") + type_name());
            }
        }
    }
    template <typename T>
        requires(std::is_arithmetic_v<T> || std::is_same_v<T, std::string>)
    operator T() const {
        return get<T>();
    }

  private:
    array_t &array_storage() {
        if (kind != value_t::array)
            throw type_error(std::string("cannot iterate over ") + type_name());
        if (!elements)
            elements = std::make_shared<array_t>();
        return *elements;
    }

    struct Parser {
        std::string const &src;
        std::size_t at;

        [[noreturn]] void fail(std::string const &why) const { throw parse_error(at + 1, why); }
        void skip_space() {
            while (at < src.size() &&
                   (src[at] == ' ' || src[at] == '\t' || src[at] == '\n' || src[at] == '\r'))
                at++;
        }
        char peek() {
            skip_space();
            if (at >= src.size())
                fail("unexpected end of input");
            return src[at];
        }
        void expect(char c) {
            if (peek() != c)
                fail(std::string("expected '") + c + "'");
            at++;
        }
        bool literal(const char *word) {
            std::size_t n = std::char_traits<char>::length(word);
            if (src.compare(at, n, word) == 0) {
                at += n;
                return true;
            }
            return false;
        }
        std::string string_body() {
            expect('"');
            std::string out;
            while (true) {
                if (at >= src.size())
                    fail("unterminated string");
                char c = src[at++];
                if (c == '"')
                    break;
                if (c == '\\') {
                    if (at >= src.size())
                        fail("unterminated escape");
                    char e = src[at++];
                    switch (e) {
                    case 'n':
                        out += '\n';
                        break;
                    case 't':
                        out += '\t';
                        break;
                    case 'r':
                        out += '\r';
                        break;
                    case 'b':
                        out += '\b';
                        break;
                    case 'f':
                        out += '\f';
                        break;
                    case 'u': {
                        if (at + 4 > src.size())
                            fail("short unicode escape");
                        unsigned code = unsigned(std::strtoul(src.substr(at, 4).c_str(), nullptr, 16));
                        at += 4;
                        if (code < 0x80) {
                            out += char(code);
                        } else if (code < 0x800) {
                            out += char(0xC0 | (code >> 6));
                            out += char(0x80 | (code & 0x3F));
                        } else {
                            out += char(0xE0 | (code >> 12));
                            out += char(0x80 | ((code >> 6) & 0x3F));
                            out += char(0x80 | (code & 0x3F));
                        }
                        break;
                    }
                    default:
                        out += e; // covers \" \\ \/
                    }
                } else {
                    out += c;
                }
            }
            return out;
        }
        json number() {
            std::size_t start = at;
            bool is_float = false;
            if (src[at] == '-')
                at++;
            while (at < src.size()) {
                char c = src[at];
                if (c >= '0' && c <= '9') {
                    at++;
                } else if (c == '.' || c == 'e' || c == 'E' || c == '+' || c == '-') {
                    is_float = true;
                    at++;
                } else {
                    break;
                }
            }
            std::string token = src.substr(start, at - start);
            if (token.empty() || token == "-")
                fail("invalid number");
            char *end = nullptr;
            if (is_float) {
                double d = std::strtod(token.c_str(), &end);
                if (*end != '\0')
                    fail("invalid number");
                return json(d);
            }
            long long i = std::strtoll(token.c_str(), &end, 10);
            if (*end != '\0')
                fail("invalid number");
            return json(std::int64_t(i));
        }
        json value() {
            char c = peek();
            if (c == '{') {
                at++;
                json obj;
                obj.kind = value_t::object;
                obj.members = std::make_shared<object_t>();
                if (peek() == '}') {
                    at++;
                    return obj;
                }
                while (true) {
                    skip_space();
                    std::string key = string_body();
                    expect(':');
                    (*obj.members)[key] = value();
                    char d = peek();
                    at++;
                    if (d == '}')
                        break;
                    if (d != ',')
                        fail("expected ',' or '}'");
                }
                return obj;
            }
            if (c == '[') {
                at++;
                json arr;
                arr.kind = value_t::array;
                arr.elements = std::make_shared<array_t>();
                if (peek() == ']') {
                    at++;
                    return arr;
                }
                while (true) {
                    arr.elements->push_back(value());
                    char d = peek();
                    at++;
                    if (d == ']')
                        break;
                    if (d != ',')
                        fail("expected ',' or ']'");
                }
                return arr;
            }
            if (c == '"')
                return json(string_body());
            if (literal("true"))
                return json(true);
            if (literal("false"))
                return json(false);
            if (literal("null"))
                return json();
            if (c == '-' || (c >= '0' && c <= '9'))
                return number();
            fail("unexpected character");
        }
    };

    value_t kind = value_t::null;
    bool boolean = false;
    std::int64_t integer = 0;
    double real = 0.0;
    std::string text;
    std::shared_ptr<object_t> members;
    std::shared_ptr<array_t> elements;
};

} // namespace nlohmann
