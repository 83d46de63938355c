// catalogue.hip — see catalogue.hpp.  Host-only: a small JSON reader (the reference links nlohmann::json, absent here),
// the catalogue mirror and SHA-256 (FIPS 180-4).
#include "catalogue.hpp"

#include "common.hpp"

#include <cctype>
#include <cstdint>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <memory>
#include <stdexcept>

namespace {

constexpr int kLogInfo = mw::LOG_INFO, kLogWarn = mw::LOG_WARN, kLogError = mw::LOG_ERROR;

// ---- JSON (RFC 8259) into a tree; numbers are kept as text (the catalogue has none that matter) ----
struct Json {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    std::string s;  // String / Number text
    std::vector<Json> a;
    std::vector<std::pair<std::string, Json>> o;
    const Json *get(const std::string &key) const {
        if (kind != Object) return nullptr;
        for (const auto &kv : o)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    bool is_string() const { return kind == String; }
};

struct JsonReader {
    const char *p, *end;
    [[noreturn]] void fail(const char *what) const { throw std::runtime_error(std::string("catalogue JSON: ") + what); }
    void ws() {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p;
    }
    void append_utf8(std::string &out, unsigned cp) {
        if (cp < 0x80) out += (char)cp;
        else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
        else if (cp < 0x10000) { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
        else { out += (char)(0xF0 | (cp >> 18)); out += (char)(0x80 | ((cp >> 12) & 0x3F)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
    }
    unsigned hex4() {
        if (end - p < 4) fail("truncated \\u escape");
        unsigned v = 0;
        for (int i = 0; i < 4; ++i) {
            const char c = *p++;
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (unsigned)(c - '0');
            else if (c >= 'a' && c <= 'f') v |= (unsigned)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (unsigned)(c - 'A' + 10);
            else fail("bad \\u escape");
        }
        return v;
    }
    std::string string() {
        if (p >= end || *p != '"') fail("string expected");
        ++p;
        std::string out;
        while (true) {
            if (p >= end) fail("unterminated string");
            const char c = *p++;
            if (c == '"') return out;
            if ((unsigned char)c < 0x20) fail("control character in string");
            if (c != '\\') {
                out += c;
                continue;
            }
            if (p >= end) fail("unterminated escape");
            const char e = *p++;
            switch (e) {
            case '"': out += '"'; break;
            case '\\': out += '\\'; break;
            case '/': out += '/'; break;
            case 'b': out += '\b'; break;
            case 'f': out += '\f'; break;
            case 'n': out += '\n'; break;
            case 'r': out += '\r'; break;
            case 't': out += '\t'; break;
            case 'u': {
                unsigned cp = hex4();
                if (cp >= 0xD800 && cp < 0xDC00 && end - p >= 6 && p[0] == '\\' && p[1] == 'u') {
                    p += 2;
                    const unsigned lo = hex4();
                    if (lo >= 0xDC00 && lo < 0xE000) cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                    else fail("bad surrogate pair");
                }
                append_utf8(out, cp);
                break;
            }
            default: fail("bad escape");
            }
        }
    }
    Json value(int depth) {
        if (depth > 64) fail("nesting too deep");
        ws();
        if (p >= end) fail("value expected");
        Json j;
        const char c = *p;
        if (c == '{') {
            ++p;
            j.kind = Json::Object;
            ws();
            if (p < end && *p == '}') { ++p; return j; }
            while (true) {
                ws();
                std::string key = string();
                ws();
                if (p >= end || *p != ':') fail("':' expected");
                ++p;
                j.o.emplace_back(std::move(key), value(depth + 1));
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; return j; }
                fail("',' or '}' expected");
            }
        }
        if (c == '[') {
            ++p;
            j.kind = Json::Array;
            ws();
            if (p < end && *p == ']') { ++p; return j; }
            while (true) {
                j.a.push_back(value(depth + 1));
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == ']') { ++p; return j; }
                fail("',' or ']' expected");
            }
        }
        if (c == '"') {
            j.kind = Json::String;
            j.s = string();
            return j;
        }
        auto word = [&](const char *w) {
            const size_t n = strlen(w);
            if ((size_t)(end - p) >= n && memcmp(p, w, n) == 0) { p += n; return true; }
            return false;
        };
        if (word("true")) { j.kind = Json::Bool; j.b = true; return j; }
        if (word("false")) { j.kind = Json::Bool; return j; }
        if (word("null")) return j;
        if (c == '-' || (c >= '0' && c <= '9')) {
            const char *s0 = p;
            while (p < end && (isdigit((unsigned char)*p) || *p == '-' || *p == '+' || *p == '.' || *p == 'e' || *p == 'E')) ++p;
            j.kind = Json::Number;
            j.s.assign(s0, p);
            return j;
        }
        fail("unexpected character");
    }
};

bool parse_model_json(const Json &model, ModelInfo &mi) {
    const Json *fn = model.get("friendly_name");
    if (!fn || !fn->is_string()) {
        mw::log_msg(kLogWarn, "Missing or invalid 'friendly_name' for a model. Skipping this model.\n");
        return false;
    }
    mi.friendly_name = fn->s;
    const Json *lf = model.get("local_folder_name");
    if (lf && lf->is_string()) mi.local_folder_name = lf->s;
    else mw::log_msg(kLogWarn, "Missing or invalid 'local_folder_name' for model: %s\n", mi.friendly_name.c_str());
    const Json *ty = model.get("type");
    if (ty && ty->is_string()) {
        if (ty->s == "MODEL_TYPE_TRANSCRIPTION") mi.type = MODEL_TYPE_TRANSCRIPTION;
        else if (ty->s == "MODEL_TYPE_TRANSLATION") mi.type = MODEL_TYPE_TRANSLATION;
        else mw::log_msg(kLogWarn, "Invalid 'type' for model: %s\n", mi.friendly_name.c_str());
    } else {
        mw::log_msg(kLogWarn, "Missing or invalid 'type' for model: %s\n", mi.friendly_name.c_str());
    }
    const Json *files = model.get("files");
    if (files && files->kind == Json::Array) {
        for (const Json &file : files->a) {
            ModelFileDownloadInfo fi;
            const Json *url = file.get("url");
            if (url && url->is_string()) fi.url = url->s;
            else mw::log_msg(kLogWarn, "Missing or invalid 'url' for a file in model: %s\n", mi.friendly_name.c_str());
            const Json *sha = file.get("sha256");
            if (sha && sha->is_string()) fi.sha256 = sha->s;
            mi.files.push_back(fi);
        }
    } else {
        mw::log_msg(kLogWarn, "Missing or invalid 'files' array for model: %s\n", mi.friendly_name.c_str());
    }
    const Json *extra = model.get("extra");
    if (extra && extra->kind == Json::Object) {
        const Json *l = extra->get("language"), *d = extra->get("description"), *s = extra->get("source");
        if (l && l->is_string()) mi.extra.language = l->s;
        if (d && d->is_string()) mi.extra.description = d->s;
        if (s && s->is_string()) mi.extra.source = s->s;
    }
    return true;
}

// ---- SHA-256 ----
struct Sha256 {
    uint32_t h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    unsigned char buf[64];
    size_t fill = 0;
    uint64_t total = 0;
    static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
    void block(const unsigned char *p) {
        static const uint32_t K[64] = {
            0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u,
            0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu,
            0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u,
            0x06ca6351u, 0x14292967u, 0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u,
            0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u, 0x19a4c116u, 0x1e376c08u,
            0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u,
            0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
        uint32_t w[64];
        for (int i = 0; i < 16; ++i)
            w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | (uint32_t)p[4 * i + 3];
        for (int i = 16; i < 64; ++i) {
            const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
            const uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; ++i) {
            const uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25), ch = (e & f) ^ (~e & g);
            const uint32_t t1 = hh + S1 + ch + K[i] + w[i];
            const uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22), mj = (a & b) ^ (a & c) ^ (b & c);
            const uint32_t t2 = S0 + mj;
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    void update(const void *data, size_t n) {
        const unsigned char *p = (const unsigned char *)data;
        total += n;
        while (n > 0) {
            const size_t take = std::min(n, (size_t)64 - fill);
            memcpy(buf + fill, p, take);
            fill += take; p += take; n -= take;
            if (fill == 64) { block(buf); fill = 0; }
        }
    }
    std::string hex() {
        const uint64_t bits = total * 8;
        const unsigned char one = 0x80, zero = 0;
        update(&one, 1);
        while (fill != 56) update(&zero, 1);
        unsigned char len[8];
        for (int i = 0; i < 8; ++i) len[i] = (unsigned char)(bits >> (56 - 8 * i));
        update(len, 8);
        static const char *dg = "0123456789abcdef";
        std::string out;
        for (int i = 0; i < 8; ++i)
            for (int s = 28; s >= 0; s -= 4) out += dg[(h[i] >> s) & 15];
        return out;
    }
};

}  // namespace

std::map<std::string, ModelInfo> load_models_info(const std::string &json_text) {
    JsonReader r{json_text.data(), json_text.data() + json_text.size()};
    const Json root = r.value(0);
    r.ws();
    if (r.p != r.end) r.fail("trailing characters");
    const Json *models = root.get("models");
    if (!models || models->kind != Json::Array) throw std::runtime_error("catalogue JSON: no \"models\" array");
    std::map<std::string, ModelInfo> out;
    for (const Json &m : models->a) {
        ModelInfo mi;
        if (parse_model_json(m, mi)) out[mi.friendly_name] = mi;
    }
    return out;
}

std::string find_model_folder(const ModelInfo &model_info, const std::string &data_models_dir, const std::string &config_models_dir) {
    if (model_info.friendly_name.empty()) {
        mw::log_msg(kLogError, "Model info is invalid. Friendly name is empty.\n");
        return "";
    }
    if (model_info.local_folder_name.empty()) {
        mw::log_msg(kLogError, "Model info is invalid. Local folder name is empty.\n");
        return "";
    }
    if (model_info.files.empty()) {
        mw::log_msg(kLogError, "Model info is invalid. Files list is empty.\n");
        return "";
    }
    namespace fs = std::filesystem;
    std::error_code ec;
    if (!data_models_dir.empty()) {
        const fs::path pth = fs::absolute(data_models_dir, ec) / model_info.local_folder_name;
        if (fs::exists(pth, ec)) return pth.string();
    }
    if (config_models_dir.empty()) {
        mw::log_msg(kLogInfo, "Config folder not set.\n");
        return "";
    }
    const fs::path pth = fs::absolute(config_models_dir, ec) / model_info.local_folder_name;
    if (fs::exists(pth, ec)) return pth.string();
    mw::log_msg(kLogInfo, "Model not found in config folder: %s\n", pth.string().c_str());
    return "";
}

std::string find_bin_file_in_folder(const std::string &model_local_folder_path) {
    namespace fs = std::filesystem;
    std::error_code ec;
    for (fs::directory_iterator it(model_local_folder_path, ec), end; !ec && it != end; it.increment(ec))
        if (it->path().extension() == ".bin") return it->path().string();
    mw::log_msg(kLogError, "Model bin file not found in folder: %s\n", model_local_folder_path.c_str());
    return "";
}

std::string find_model_bin_file(const ModelInfo &model_info, const std::string &data_models_dir, const std::string &config_models_dir) {
    const std::string folder = find_model_folder(model_info, data_models_dir, config_models_dir);
    return folder.empty() ? "" : find_bin_file_in_folder(folder);
}

std::string sha256_hex(const void *data, size_t n) {
    Sha256 s;
    s.update(data, n);
    return s.hex();
}

std::string sha256_file(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("sha256: cannot open " + path);
    Sha256 s;
    std::unique_ptr<char[]> buf(new char[1 << 20]);
    while (f) {
        f.read(buf.get(), 1 << 20);
        const std::streamsize got = f.gcount();
        if (got > 0) s.update(buf.get(), (size_t)got);
    }
    if (f.bad()) throw std::runtime_error("sha256: read error on " + path);
    return s.hex();
}

int verify_model_file(const ModelInfo &model_info, const std::string &bin_path) {
    if (model_info.files.empty() || model_info.files[0].sha256.empty()) return 2;
    std::string want = model_info.files[0].sha256;
    for (char &c : want) c = (char)tolower((unsigned char)c);
    return sha256_file(bin_path) == want ? 0 : 1;
}
