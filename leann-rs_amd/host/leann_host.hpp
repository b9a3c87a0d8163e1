// leann_host.hpp — C++ host-side mirror of leann-rs's index layer above the C ABI
// (the reference is Rust; no Rust toolchain exists in this image).
// Same names, argument meaning and error behaviour as the reference for the path:
//     IndexMeta            src/index/meta.rs:9-58
//     PassageStore         src/index/passages.rs:11-105
//     tokenize / Bm25Scorer / hybrid_rerank   src/index/bm25.rs:9-170   (f32, op for op)
//     MetadataFilter       src/index/filter.rs (mini-language evaluated on passage metadata)
//     find_index           src/index/locate.rs:11-36
//     SearchOptions / SearchResult / IndexSearcher   src/index/searcher.rs:15-257
//     RecomputeSearcher    src/index/recompute.rs:17-139   (arithmetic on the GPU: leann_scan_topk_device)
// Vector search itself always goes through include/leann_backend.h (HIP kernels); nothing here
// computes distances on the CPU.
#pragma once
#include "../../include/leann_backend.h"
#include "json.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <optional>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <sys/stat.h>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace leann {

struct Error : std::runtime_error { using std::runtime_error::runtime_error; };
inline void check(int rc) { if (rc != 0) throw Error(leann_last_error()); }

inline bool file_exists(const std::string &p) { struct stat st; return ::stat(p.c_str(), &st) == 0; }
inline std::string read_file(const std::string &p) {
    std::ifstream f(p, std::ios::binary);
    if (!f) throw Error("No such file or directory: " + p);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}
// Path::with_extension on ".../documents.leann": replace the text after the last '.' of the file name
inline std::string with_extension(const std::string &stem, const std::string &ext) {
    size_t slash = stem.find_last_of('/'), dot = stem.find_last_of('.');
    bool has = dot != std::string::npos && (slash == std::string::npos || dot > slash + 1);
    return (has ? stem.substr(0, dot) : stem) + "." + ext;
}

// ---- src/index/meta.rs ---------------------------------------------------------------------------
struct IndexMeta {
    std::string version, backend_name, embedding_model, embedding_mode;
    size_t dimensions = 0, passage_count = 0;
    lj::Value backend_kwargs, embedding_options; // optional
    bool is_recompute = false, is_pruned = false;

    static IndexMeta load(const std::string &path) {
        lj::Value v = lj::parse(read_file(path));
        auto need = [&](const char *k) -> const lj::Value & {
            const lj::Value *x = v.get(k);
            if (!x) throw Error(std::string("missing field `") + k + "`");
            return *x;
        };
        IndexMeta m;
        m.version = need("version").s;
        m.backend_name = need("backend_name").s;
        m.embedding_model = need("embedding_model").s;
        m.embedding_mode = need("embedding_mode").s;
        m.dimensions = (size_t)need("dimensions").as_f64();
        m.passage_count = (size_t)need("passage_count").as_f64();
        if (auto *x = v.get("backend_kwargs")) m.backend_kwargs = *x;
        if (auto *x = v.get("embedding_options")) m.embedding_options = *x;
        if (auto *x = v.get("is_recompute")) m.is_recompute = x->b;   // #[serde(default)]
        if (auto *x = v.get("is_pruned")) m.is_pruned = x->b;
        return m;
    }
    void save(const std::string &path) const {
        lj::Value v = lj::Value::object();
        v["version"] = lj::Value::string(version);
        v["backend_name"] = lj::Value::string(backend_name);
        v["embedding_model"] = lj::Value::string(embedding_model);
        v["embedding_mode"] = lj::Value::string(embedding_mode);
        v["dimensions"] = lj::Value::integer((int64_t)dimensions);
        v["passage_count"] = lj::Value::integer((int64_t)passage_count);
        if (!backend_kwargs.is_null()) v["backend_kwargs"] = backend_kwargs; // skip_serializing_if None
        if (!embedding_options.is_null()) v["embedding_options"] = embedding_options;
        v["is_recompute"] = lj::Value::boolean(is_recompute);
        v["is_pruned"] = lj::Value::boolean(is_pruned);
        std::ofstream(path) << lj::to_string_pretty(v);
    }
};

// ---- src/index/passages.rs -------------------------------------------------------------------------
struct Passage {
    std::string id, text;
    lj::Value metadata;
};
class PassageStore {
  public:
    static PassageStore open(const std::string &base_path) {
        PassageStore s;
        s.jsonl_path_ = with_extension(base_path, "passages.jsonl");
        lj::Value idx = lj::parse(read_file(with_extension(base_path, "passages.idx.json")));
        if (!idx.is_object()) throw Error("passages.idx.json: expected an object");
        for (auto &kv : *idx.o) s.offsets_[kv.first] = (uint64_t)kv.second.as_f64();
        return s;
    }
    Passage get(const std::string &id) const { // one open + seek + one line, like passages.rs:90-105
        auto it = offsets_.find(id);
        if (it == offsets_.end()) throw Error("Passage not found: " + id);
        std::ifstream f(jsonl_path_, std::ios::binary);
        if (!f) throw Error("cannot open " + jsonl_path_);
        f.seekg((std::streamoff)it->second);
        std::string line;
        std::getline(f, line);
        lj::Value v = lj::parse(line);
        Passage p;
        if (auto *x = v.get("id")) p.id = x->s;
        if (auto *x = v.get("text")) p.text = x->s;
        if (auto *x = v.get("metadata")) p.metadata = *x;
        return p;
    }
    std::vector<std::string> ids() const {
        std::vector<std::string> r;
        for (auto &kv : offsets_) r.push_back(kv.first);
        return r;
    }
    size_t len() const { return offsets_.size(); }

  private:
    std::unordered_map<std::string, uint64_t> offsets_;
    std::string jsonl_path_;
};
class PassageStoreWriter { // passages.rs:120-158
  public:
    explicit PassageStoreWriter(const std::string &base_path)
        : idx_path_(with_extension(base_path, "passages.idx.json")), out_(with_extension(base_path, "passages.jsonl"), std::ios::binary) {}
    void add(const Passage &p) {
        lj::Value v = lj::Value::object();
        v["id"] = lj::Value::string(p.id);
        v["text"] = lj::Value::string(p.text);
        v["metadata"] = p.metadata;
        std::string line = lj::to_string(v);
        offsets_[p.id] = cur_;
        out_ << line << '\n';
        cur_ += line.size() + 1;
    }
    void finish() {
        out_.flush();
        lj::Value idx = lj::Value::object();
        for (auto &kv : offsets_) idx[kv.first] = lj::Value::integer((int64_t)kv.second);
        std::ofstream(idx_path_) << lj::to_string(idx);
    }

  private:
    std::string idx_path_;
    std::ofstream out_;
    std::map<std::string, uint64_t> offsets_;
    uint64_t cur_ = 0;
};

// ---- src/index/bm25.rs -----------------------------------------------------------------------------
inline std::vector<std::string> tokenize(const std::string &text) { // [a-zA-Z0-9]+, lowercase, len > 1
    std::vector<std::string> out;
    std::string cur;
    auto flush = [&] { if (cur.size() > 1) out.push_back(cur); cur.clear(); };
    for (unsigned char c : text) {
        if ((c >= 'a' && c <= 'z') || (c >= '0' && c <= '9')) cur += (char)c;
        else if (c >= 'A' && c <= 'Z') cur += (char)(c - 'A' + 'a');
        else flush();
    }
    flush();
    return out;
}

class Bm25Scorer {
  public:
    static constexpr float K1 = 1.2f, B = 0.75f; // bm25.rs:9-10
    static Bm25Scorer build(const std::vector<std::string> &documents) {
        Bm25Scorer s;
        s.num_docs_ = documents.size();
        size_t total = 0;
        for (auto &doc : documents) {
            auto toks = tokenize(doc);
            s.doc_lengths_.push_back(toks.size());
            total += toks.size();
            std::unordered_map<std::string, size_t> tf;
            for (auto &t : toks) if (tf[t]++ == 0) s.doc_freq_[t]++;
            s.term_freqs_.push_back(std::move(tf));
        }
        s.avg_doc_len_ = s.num_docs_ > 0 ? (float)total / (float)s.num_docs_ : 1.0f;
        return s;
    }
    std::vector<float> score_query(const std::string &query) const {
        std::vector<float> scores(num_docs_, 0.0f);
        for (auto &token : tokenize(query)) {
            auto it = doc_freq_.find(token);
            float df = it == doc_freq_.end() ? 0.0f : (float)it->second;
            if (df == 0.0f) continue;
            float idf = std::log(((float)num_docs_ - df + 0.5f) / (df + 0.5f) + 1.0f); // :88
            for (size_t doc = 0; doc < num_docs_; doc++) {
                auto jt = term_freqs_[doc].find(token);
                float tf = jt == term_freqs_[doc].end() ? 0.0f : (float)jt->second;
                if (tf == 0.0f) continue;
                float doc_len = (float)doc_lengths_[doc];
                float norm = 1.0f - B + B * (doc_len / avg_doc_len_);     // :97
                float score = idf * (tf * (K1 + 1.0f)) / (tf + K1 * norm); // :100
                scores[doc] += score;
            }
        }
        return scores;
    }
    std::vector<std::pair<size_t, float>> search(const std::string &query, size_t top_k) const {
        auto scores = score_query(query);
        std::vector<std::pair<size_t, float>> scored;
        for (size_t i = 0; i < scores.size(); i++)
            if (scores[i] > 0.0f) scored.emplace_back(i, scores[i]);
        std::stable_sort(scored.begin(), scored.end(), [](auto &a, auto &b) { return b.second < a.second; });
        if (scored.size() > top_k) scored.resize(top_k);
        return scored;
    }

  private:
    std::unordered_map<std::string, size_t> doc_freq_;
    size_t num_docs_ = 0;
    float avg_doc_len_ = 1.0f;
    std::vector<size_t> doc_lengths_;
    std::vector<std::unordered_map<std::string, size_t>> term_freqs_;
};

// bm25.rs:135-170 — min-max normalise both lists, alpha blend, stable sort descending
inline std::vector<std::pair<size_t, float>> hybrid_rerank(const std::vector<std::pair<size_t, float>> &vector_results,
                                                           const std::vector<float> &bm25_scores, float alpha) {
    float max_v = -INFINITY, min_v = INFINITY;
    for (auto &r : vector_results) { max_v = std::fmax(max_v, r.second); min_v = std::fmin(min_v, r.second); }
    float vrange = std::fmax(max_v - min_v, 1e-6f);
    float max_b = -INFINITY, min_b = INFINITY;
    for (float b : bm25_scores) { max_b = std::fmax(max_b, b); min_b = std::fmin(min_b, b); }
    float brange = std::fmax(max_b - min_b, 1e-6f);
    std::vector<std::pair<size_t, float>> out;
    out.reserve(vector_results.size());
    for (auto &r : vector_results) {
        float norm_vec = (r.second - min_v) / vrange;
        float bm = r.first < bm25_scores.size() ? bm25_scores[r.first] : 0.0f;
        float norm_b = (bm - min_b) / brange;
        float t1 = alpha * norm_vec, t2 = (1.0f - alpha) * norm_b;
        out.emplace_back(r.first, t1 + t2);
    }
    std::stable_sort(out.begin(), out.end(), [](auto &a, auto &b) { return b.second < a.second; });
    return out;
}

// ---- src/index/filter.rs (behavioural mirror of the mini-language) -------------------------------------
class MetadataFilter {
  public:
    enum Op { Eq, Ne, Gt, Gte, Lt, Lte, In, NotIn, Contains, StartsWith, EndsWith, Exists };
    static std::optional<MetadataFilter> parse(const std::string &text) {
        std::string s = trim(text);
        if (s.find(" OR ") != std::string::npos) return combine(split(s, " OR "), false, false);
        bool has_and = s.find(" AND ") != std::string::npos;
        if (has_and) return combine(split(s, " AND "), true, true);
        auto parts = split_commas(s);
        if (parts.size() > 1) return combine(parts, true, true);
        return single(s);
    }
    bool matches(const lj::Value &metadata) const {
        if (kind_ == And) { for (auto &c : children_) if (!c.matches(metadata)) return false; return true; }
        if (kind_ == Or) { for (auto &c : children_) if (c.matches(metadata)) return true; return false; }
        const lj::Value *fv = &metadata;
        {
            size_t start = 0;
            for (;;) {
                size_t dot = field_.find('.', start);
                std::string part = field_.substr(start, dot == std::string::npos ? std::string::npos : dot - start);
                fv = fv->get(part);
                if (!fv || dot == std::string::npos) break;
                start = dot + 1;
            }
        }
        auto str_test = [&](auto pred) { return fv && fv->is_string() && value_.is_string() ? pred(fv->s, value_.s) : (fv && fv->is_string() && pred(fv->s, std::string())); };
        switch (op_) {
            case Exists: return fv != nullptr;
            case Eq: return fv && equal(*fv, value_);
            case Ne: return !fv || !equal(*fv, value_);
            case Gt: return fv && compare(*fv, value_) > 0;
            case Gte: return fv && compare(*fv, value_) >= 0;
            case Lt: return fv && compare(*fv, value_) < 0;
            case Lte: return fv && compare(*fv, value_) <= 0;
            case In: if (!fv) return false; for (auto &x : *value_.a) if (equal(*fv, x)) return true; return false;
            case NotIn: if (!fv) return true; for (auto &x : *value_.a) if (equal(*fv, x)) return false; return true;
            case Contains: return str_test([](const std::string &h, const std::string &n) { return h.find(n) != std::string::npos; });
            case StartsWith: return str_test([](const std::string &h, const std::string &n) { return h.compare(0, n.size(), n) == 0; });
            case EndsWith: return str_test([](const std::string &h, const std::string &n) { return h.size() >= n.size() && h.compare(h.size() - n.size(), n.size(), n) == 0; });
        }
        return false;
    }

  private:
    enum Kind { Cond, And, Or } kind_ = Cond;
    std::string field_;
    Op op_ = Eq;
    lj::Value value_;
    std::vector<MetadataFilter> children_;

    static std::string trim(const std::string &s) {
        size_t a = s.find_first_not_of(" \t\n\r"), b = s.find_last_not_of(" \t\n\r");
        return a == std::string::npos ? "" : s.substr(a, b - a + 1);
    }
    static std::vector<std::string> split(const std::string &s, const std::string &sep) {
        std::vector<std::string> out;
        size_t pos = 0, f;
        while ((f = s.find(sep, pos)) != std::string::npos) { out.push_back(s.substr(pos, f - pos)); pos = f + sep.size(); }
        out.push_back(s.substr(pos));
        return out;
    }
    static std::vector<std::string> split_commas(const std::string &s) { // commas outside [...]
        std::vector<std::string> out;
        std::string cur;
        int depth = 0;
        for (char c : s) {
            if (c == '[') depth++;
            if (c == ']') depth--;
            if (c == ',' && depth == 0) { out.push_back(cur); cur.clear(); } else cur += c;
        }
        if (!cur.empty()) out.push_back(cur);
        return out;
    }
    static std::optional<MetadataFilter> combine(const std::vector<std::string> &parts, bool is_and, bool singles) {
        std::vector<MetadataFilter> fs;
        for (auto &p : parts) {
            auto f = singles ? single(trim(p)) : parse(trim(p));
            if (f) fs.push_back(*f);
        }
        if (fs.size() > 1) { MetadataFilter m; m.kind_ = is_and ? And : Or; m.children_ = std::move(fs); return m; }
        if (fs.size() == 1) return fs[0];
        return std::nullopt;
    }
    static lj::Value parse_value(const std::string &s) {
        char *end = nullptr;
        if (!s.empty()) {
            long long iv = strtoll(s.c_str(), &end, 10);
            if (end && *end == 0 && (isdigit((unsigned char)s[0]) || s[0] == '-' || s[0] == '+')) return lj::Value::integer(iv);
            double dv = strtod(s.c_str(), &end);
            if (end && *end == 0 && std::isfinite(dv) && (isdigit((unsigned char)s[0]) || s[0] == '-' || s[0] == '+' || s[0] == '.')) return lj::Value::number(dv);
        }
        if (s == "true") return lj::Value::boolean(true);
        if (s == "false") return lj::Value::boolean(false);
        return lj::Value::string(s);
    }
    static MetadataFilter cond(std::string field, Op op, lj::Value v) {
        MetadataFilter m;
        m.field_ = std::move(field);
        m.op_ = op;
        m.value_ = std::move(v);
        return m;
    }
    static std::optional<MetadataFilter> list_op(const std::string &s, const std::string &kw, Op op) {
        size_t idx = s.find(kw);
        if (idx == std::string::npos) return std::nullopt;
        std::string rest = s.substr(idx + kw.size());
        size_t end = rest.find(']');
        if (end == std::string::npos) return std::nullopt;
        lj::Value arr = lj::Value::array();
        for (auto &v : split(rest.substr(0, end), ",")) arr.a->push_back(parse_value(trim(v)));
        return cond(trim(s.substr(0, idx)), op, arr);
    }
    static std::optional<MetadataFilter> single(const std::string &text) {
        std::string s = trim(text);
        if (s.empty()) return std::nullopt;
        if (s.back() == '?') return cond(s.substr(0, s.size() - 1), Exists, lj::Value());
        if (auto f = list_op(s, " in [", In)) return f;
        if (auto f = list_op(s, " not_in [", NotIn)) return f;
        auto two = [&](const std::string &sep, Op op, bool as_string) -> std::optional<MetadataFilter> {
            size_t p = s.find(sep);
            if (p == std::string::npos) return std::nullopt;
            std::string rhs = s.substr(p + sep.size());
            return cond(s.substr(0, p), op, as_string ? lj::Value::string(rhs) : parse_value(rhs));
        };
        if (s.find('~') != std::string::npos) return two("~", Contains, true);
        if (s.find('^') != std::string::npos && s.find(">=") == std::string::npos) return two("^", StartsWith, true);
        if (s.find('$') != std::string::npos) return two("$", EndsWith, true);
        if (s.find("!=") != std::string::npos) return two("!=", Ne, false);
        if (s.find(">=") != std::string::npos) return two(">=", Gte, false);
        if (s.find("<=") != std::string::npos) return two("<=", Lte, false);
        if (s.find('>') != std::string::npos) return two(">", Gt, false);
        if (s.find('<') != std::string::npos) return two("<", Lt, false);
        size_t p = s.find('=');
        if (p == std::string::npos) p = s.find(':');
        if (p == std::string::npos) return std::nullopt;
        std::string field = s.substr(0, p), value = s.substr(p + 1);
        if (value.find('*') != std::string::npos) { // glob forms
            bool st = value.front() == '*', en = value.back() == '*';
            if (st && en && value.size() > 2) return cond(field, Contains, lj::Value::string(value.substr(1, value.size() - 2)));
            if (st) return cond(field, EndsWith, lj::Value::string(value.substr(1)));
            if (en) return cond(field, StartsWith, lj::Value::string(value.substr(0, value.size() - 1)));
        }
        return cond(field, Eq, parse_value(value));
    }
    static bool equal(const lj::Value &a, const lj::Value &b) {
        if (a.is_string() && b.is_string()) return a.s == b.s;
        if (a.is_number() && b.is_number()) return std::fabs(a.as_f64() - b.as_f64()) < std::numeric_limits<double>::epsilon();
        if (a.kind == lj::Value::Bool && b.kind == lj::Value::Bool) return a.b == b.b;
        return a.is_null() && b.is_null();
    }
    static int compare(const lj::Value &a, const lj::Value &b) {
        if (a.is_number() && b.is_number()) return a.as_f64() < b.as_f64() ? -1 : (a.as_f64() > b.as_f64() ? 1 : 0);
        if (a.is_string() && b.is_string()) return a.s < b.s ? -1 : (a.s > b.s ? 1 : 0);
        return 0;
    }
};

// ---- src/index/locate.rs ---------------------------------------------------------------------------
inline std::string find_index(const std::string &name) {
    std::string local = ".leann/indexes/" + name;
    if (file_exists(local)) return local;
    if (!name.empty() && name[0] == '/' && file_exists(name)) return name;
    if (const char *home = getenv("HOME")) {
        std::string global = std::string(home) + "/.leann/indexes/" + name;
        if (file_exists(global)) return global;
    }
    throw Error("Index '" + name + "' not found. Run 'leann list' to see available indexes.");
}

// ---- src/index/searcher.rs ---------------------------------------------------------------------------
struct SearchResult {
    std::string id;
    float score = 0.f;
    std::string text;
    lj::Value metadata;
};
struct SearchOptions {
    size_t top_k = 5, complexity = 64;
    std::optional<MetadataFilter> filter;
    bool hybrid = false;
    float hybrid_alpha = 0.7f; // searcher.rs:47
    std::optional<std::string> query_text;
    // Additive: evaluate the metadata filter INSIDE the graph traversal (allow-bitmap over
    // positions, leann_backend_search_filtered) instead of over-fetching 5*top_k and post-filtering (:129-133,
    // :190-194).  `filter_key` names the filter for the bitmap cache (the CLI passes the filter text).
    bool device_filter = false;
    std::string filter_key;
    // the reference hands the backend's DISTANCES (1 - dot, lower = better) to hybrid_rerank, which treats a larger
    // vector score as better (bm25.rs:159, sort desc :168) — in hybrid mode the WORST ANN hits rank highest on the vector term.
    // true (default) reproduces that faithfully; false is the corrected mode: the ANN hits enter the blend as similarities 1 - dist
    // (BM25-only hits keep the 0.0 of searcher.rs:160-165).  `leann search --compat-polarity true|false`.
    bool compat_polarity = true;
    SearchOptions(size_t k, size_t c) : top_k(k), complexity(c) {}
    SearchOptions &with_device_filter(std::string key) { device_filter = true; filter_key = std::move(key); return *this; }
    SearchOptions &with_filter(MetadataFilter f) { filter = std::move(f); return *this; }
    SearchOptions &with_hybrid(std::string q, float alpha) { hybrid = true; hybrid_alpha = alpha; query_text = std::move(q); return *this; }
    SearchOptions &with_compat_polarity(bool on) { compat_polarity = on; return *this; }
};

inline std::vector<std::string> read_id_map(const std::string &index_path, const PassageStore &passages) {
    std::string ids_path = with_extension(index_path, "ids.txt"); // searcher.rs:83-92
    std::vector<std::string> ids;
    if (file_exists(ids_path)) {
        std::istringstream ss(read_file(ids_path));
        for (std::string line; std::getline(ss, line);) {
            if (!line.empty() && line.back() == '\r') line.pop_back();
            ids.push_back(line);
        }
    } else {
        ids = passages.ids();
    }
    return ids;
}

class IndexSearcher {
  public:
    static IndexSearcher load(const std::string &index_path, const IndexMeta &meta, const char *device = "0") {
        IndexSearcher s;
        s.passages_ = PassageStore::open(index_path);
        s.id_map_ = read_id_map(index_path, s.passages_);
        int backend;
        if (meta.backend_name == "hnsw") backend = LEANN_BACKEND_HNSW;
        else if (meta.backend_name == "diskann") backend = LEANN_BACKEND_DISKANN;
        else throw Error("Unknown backend: " + meta.backend_name); // searcher.rs:98
        leann_backend *h = nullptr;
        check(leann_backend_open(index_path.c_str(), backend, meta.dimensions, device, &h));
        s.backend_.reset(h, leann_backend_close);
        return s;
    }
    // passages + id map only, no backend: assemble_results on recorded backend output (CPU tests)
    static IndexSearcher load_passages_only(const std::string &index_path) {
        IndexSearcher s;
        s.passages_ = PassageStore::open(index_path);
        s.id_map_ = read_id_map(index_path, s.passages_);
        return s;
    }
    std::vector<SearchResult> search(const std::vector<float> &q, size_t top_k, size_t complexity) const {
        return search_with_options(q, SearchOptions(top_k, complexity));
    }
    // searcher.rs:123-210
    std::vector<SearchResult> search_with_options(const std::vector<float> &query_embedding, const SearchOptions &opts) const {
        size_t fetch_k = (opts.filter || opts.hybrid) ? opts.top_k * 5 : opts.top_k; // :129-133
        std::vector<uint64_t> keys(std::max<size_t>(fetch_k, 1));
        std::vector<float> dists(std::max<size_t>(fetch_k, 1));
        size_t n = 0;
        if (query_embedding.size() != leann_backend_dims(backend_.get()))
            throw Error("query embedding has " + std::to_string(query_embedding.size()) + " dimensions, index has " +
                        std::to_string(leann_backend_dims(backend_.get())));
        if (opts.filter && opts.device_filter) {
            // every returned position already passes the filter, so no over-fetch is needed for it
            if (!opts.hybrid) { fetch_k = opts.top_k; }
            // The filter lives on the device (bitmap + compacted list of allowed positions), registered once per distinct filter.
            const RegisteredFilter rf = device_filter(*opts.filter, opts.filter_key);
            // The walk answers from the pool of evaluated nodes (20-33 x complexity for M = 16-32): widen it so that the pool is
            // expected to hold >= 8 x fetch_k allowed passages (selectivity s: complexity >= 8 fetch_k / (20 s)), capped at 1024.
            size_t complexity = opts.complexity;
            const size_t n_rows = std::max<size_t>(leann_backend_len(backend_.get()), 1);
            if (rf.allowed > 0) {
                const double s_sel = (double)rf.allowed / (double)n_rows;
                complexity = std::min<size_t>(1024, std::max<size_t>(complexity, (size_t)std::ceil(8.0 * (double)fetch_k / (20.0 * s_sel))));
            }
            // mode 2: the library answers selective filters (<= 5 % of the rows, or <= 64k rows) exactly — the allowed rows are scanned
            // on the device — and walks otherwise.  One query at a time, 10M x 768 rows (scripts/filter_latency.py): 3 % allowed: 0.49 ms
            // exact against 0.80 ms walking at recall 0.92; 1 %: 0.32 ms against 1.8 ms at 0.84; 10 %: 1.1 ms against 0.46 ms at 0.89.
            uint32_t cnt = 0;
            check(leann_backend_search_filter_batch(backend_.get(), query_embedding.data(), 1, fetch_k, complexity, rf.handle.get(), 2,
                                                    keys.data(), dists.data(), &cnt));
            n = cnt;
        } else {
            check(leann_backend_search(backend_.get(), query_embedding.data(), fetch_k, opts.complexity, keys.data(), dists.data(), &n));
        }
        std::vector<std::pair<size_t, float>> vector_results;
        for (size_t i = 0; i < n; i++) vector_results.emplace_back((size_t)keys[i], dists[i]); // score = backend distance (N1)
        return assemble_results(std::move(vector_results), opts, fetch_k);
    }
    // Everything of search_with_options after the backend call (:146-207): hybrid merge, id map, passage fetch, post-filter, top_k cut.
    // Public so that the CPU suite can drive it with recorded backend output (tests/golden/searcher_cases.json, host_selftest).
    std::vector<SearchResult> assemble_results(std::vector<std::pair<size_t, float>> vector_results, const SearchOptions &opts,
                                               size_t fetch_k) const {
        if (opts.hybrid && opts.query_text) { // :146-169
            // The reference re-reads every passage and rebuilds the BM25 tables per query (:149-151, :213-224);
            // the index is immutable while open, so the tables are built once and kept —
            // scores are identical.
            if (!opts.compat_polarity)
                for (auto &r : vector_results) r.second = 1.0f - r.second; // corrected polarity (N1): similarity, larger = better
            const Bm25Scorer &scorer = bm25();
            auto bm25_scores = scorer.score_query(*opts.query_text);
            auto bm25_top = scorer.search(*opts.query_text, fetch_k);
            std::unordered_set<size_t> have;
            for (auto &r : vector_results) have.insert(r.first);
            for (auto &b : bm25_top)
                if (!have.count(b.first)) vector_results.emplace_back(b.first, 0.0f);
            vector_results = hybrid_rerank(vector_results, bm25_scores, opts.hybrid_alpha);
        }
        std::vector<SearchResult> results;
        for (auto &r : vector_results) {
            if (results.size() >= opts.top_k) break;
            std::string id = r.first < id_map_.size() ? id_map_[r.first] : std::to_string(r.first); // :180-184
            try {
                Passage p = passages_.get(id);
                if (opts.filter && !opts.filter->matches(p.metadata)) continue;
                results.push_back({id, r.second, p.text, p.metadata});
            } catch (const std::exception &e) {
                fprintf(stderr, "WARN Failed to load passage %s: %s\n", id.c_str(), e.what());
            }
        }
        return results;
    }
    std::vector<std::string> get_all_texts() const { // :213-224
        std::vector<std::string> texts;
        texts.reserve(id_map_.size());
        for (auto &id : id_map_) {
            try { texts.push_back(passages_.get(id).text); } catch (...) { texts.emplace_back(); }
        }
        return texts;
    }
    std::vector<std::string> bm25_search(const std::string &query, size_t top_k) const { // :228-246
        const Bm25Scorer &scorer = bm25();
        std::vector<std::string> texts;
        for (auto &r : scorer.search(query, top_k))
            if (r.first < id_map_.size()) {
                try { texts.push_back(passages_.get(id_map_[r.first]).text); } catch (...) {}
            }
        return texts;
    }
    size_t len() const { return leann_backend_len(backend_.get()); }
    bool is_empty() const { return len() == 0; }

  private:
    const Bm25Scorer &bm25() const {
        std::lock_guard<std::mutex> lk(*bm25_mu_);
        if (!bm25_) bm25_ = std::make_shared<Bm25Scorer>(Bm25Scorer::build(get_all_texts()));
        return *bm25_;
    }
    // one pass over the passage metadata per distinct filter (the index is immutable while open); bit i = position i.  The bitmap is
    // registered on the device once (leann_backend_filter_create) and reused by every query under the same filter.
    struct RegisteredFilter {
        std::shared_ptr<leann_filter> handle;
        size_t allowed = 0;
    };
    RegisteredFilter device_filter(const MetadataFilter &f, const std::string &key) const { // by value: keeps the handle alive past an eviction
        std::lock_guard<std::mutex> lk(*bm25_mu_);
        if (!key.empty()) {
            auto it = filters_->find(key);
            if (it != filters_->end()) return it->second;
        }
        const size_t n = leann_backend_len(backend_.get());
        std::vector<uint8_t> bm((n + 7) / 8 + 1, 0);
        for (size_t i = 0; i < n; i++) {
            std::string id = i < id_map_.size() ? id_map_[i] : std::to_string(i);
            try {
                if (f.matches(passages_.get(id).metadata)) bm[i >> 3] |= (uint8_t)(1u << (i & 7));
            } catch (...) {} // unreadable passage: never returned (the post-filter path skips it with a warning too)
        }
        leann_filter *raw = nullptr;
        check(leann_backend_filter_create(backend_.get(), bm.data(), &raw));
        RegisteredFilter rf;
        rf.handle = std::shared_ptr<leann_filter>(raw, [](leann_filter *p) { leann_backend_filter_free(p); });
        rf.allowed = leann_backend_filter_count(raw);
        if (filters_->size() >= 16) filters_->clear();
        return (*filters_)[key.empty() ? std::string("\x01anon") : key] = std::move(rf);
    }
    PassageStore passages_;
    std::shared_ptr<leann_backend> backend_;
    std::vector<std::string> id_map_;
    std::shared_ptr<std::map<std::string, RegisteredFilter>> filters_ = std::make_shared<std::map<std::string, RegisteredFilter>>();
    mutable std::shared_ptr<Bm25Scorer> bm25_;
    std::shared_ptr<std::mutex> bm25_mu_ = std::make_shared<std::mutex>();
};

// ---- embeddings: the thing RecomputeSearcher calls (src/embedding/mod.rs:112-143) ------------------------
// Every real provider of the reference needs the network (OpenAI / Ollama / Gemini HTTP, HF hub);
// offline the host offers a deterministic "synthetic" text embedder (hashed token directions,
// L2-normalised like every registry model, src/embedding/models.rs:39-120) and raw vector files.
inline uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
inline std::vector<float> synthetic_embed(const std::string &text, size_t dims) {
    std::vector<float> v(dims, 0.0f);
    auto toks = tokenize(text);
    if (toks.empty()) toks.push_back(text);
    for (auto &t : toks) {
        uint64_t h = 0xcbf29ce484222325ull;
        for (unsigned char c : t) h = (h ^ c) * 0x100000001b3ull;
        for (size_t j = 0; j < dims; j++) {
            uint64_t r = mix64(mix64(h) ^ (j * 0xA24BAED4963EE407ull));
            int32_t s = (int32_t)((r & 0xFFFF) + ((r >> 16) & 0xFFFF) + ((r >> 32) & 0xFFFF) + (r >> 48));
            v[j] += (float)(s - 131070) * 2.6428996e-05f;
        }
    }
    float ss = 0.0f;
    for (float x : v) ss += x * x;
    float nrm = std::sqrt(ss);
    if (nrm < 1e-12f) nrm = 1e-12f; // candle.rs:218-225
    for (float &x : v) x /= nrm;
    return v;
}
// "synthetic-linear": the recompute path's device-resident provider fed from text — compact per-passage features
// + a dense layer + L2 normalisation (the tail of a real encoder, candle.rs:165,218-225):
//     f = signed hashed bag of tokens [LINEAR_H]  (small integers: exact in bf16),   W [LINEAR_H x dims] bf16, seeded Gaussian,
//     embedding = l2_normalize(W^T f).
// An index built with `--recompute-graph` stores the graph and the features (520 B per passage), never the embeddings; the host
// only embeds the QUERY text (this function), distances are recomputed on the device during the walk.
constexpr size_t LINEAR_H = 256;
inline uint16_t bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
inline float bf16_to_f32(uint16_t b) {
    uint32_t u = (uint32_t)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
inline std::vector<uint16_t> linear_features(const std::string &text) {
    std::vector<float> f(LINEAR_H, 0.0f);
    auto toks = tokenize(text);
    if (toks.empty()) toks.push_back(text);
    for (auto &t : toks) {
        uint64_t h = 0xcbf29ce484222325ull;
        for (unsigned char c : t) h = (h ^ c) * 0x100000001b3ull;
        for (int rep = 0; rep < 4; rep++) { // four signed buckets per token: fewer collisions between distinct tokens
            const uint64_t r = mix64(h + (uint64_t)rep * 0x9E3779B97F4A7C15ull);
            f[r % LINEAR_H] += (r >> 63) ? 1.0f : -1.0f;
        }
    }
    std::vector<uint16_t> out(LINEAR_H);
    for (size_t k = 0; k < LINEAR_H; k++) out[k] = bf16_rne(f[k]);
    return out;
}
inline std::vector<uint16_t> linear_weights(size_t dims) { // [LINEAR_H x dims] bf16, row-major
    std::vector<uint16_t> W(LINEAR_H * dims);
    for (size_t k = 0; k < LINEAR_H; k++)
        for (size_t j = 0; j < dims; j++) {
            const uint64_t r = mix64(mix64(0x4C494E454152ull ^ (k * 0xD1342543DE82EF95ull)) ^ (j * 0xA24BAED4963EE407ull));
            const int32_t sgn = (int32_t)((r & 0xFFFF) + ((r >> 16) & 0xFFFF) + ((r >> 32) & 0xFFFF) + (r >> 48));
            W[k * dims + j] = bf16_rne((float)(sgn - 131070) * 2.6428996e-05f);
        }
    return W;
}
inline std::vector<float> linear_embed(const std::string &text, const std::vector<uint16_t> &W, size_t dims) {
    const auto f = linear_features(text);
    std::vector<float> e(dims, 0.0f);
    for (size_t k = 0; k < LINEAR_H; k++) {
        const float fk = bf16_to_f32(f[k]);
        if (fk == 0.0f) continue;
        for (size_t j = 0; j < dims; j++) e[j] = std::fma(bf16_to_f32(W[k * dims + j]), fk, e[j]);
    }
    float ss = 0.0f;
    for (float x : e) ss += x * x;
    float nrm = std::sqrt(ss);
    if (nrm < 1e-12f) nrm = 1e-12f;
    for (float &x : e) x /= nrm;
    return e;
}

class EmbeddingProvider {
  public:
    EmbeddingProvider(std::string mode, size_t dims) : mode_(std::move(mode)), dims_(dims) {
        if (mode_ != "synthetic" && mode_ != "synthetic-linear")
            throw Error("Embedding mode '" + mode_ + "' needs a network service that is unavailable in this build; "
                        "use --embedding-mode synthetic or --query-vector-file");
        if (mode_ == "synthetic-linear") weights_ = linear_weights(dims_);
    }
    std::vector<std::vector<float>> embed(const std::vector<std::string> &texts) const {
        std::vector<std::vector<float>> out;
        for (auto &t : texts) out.push_back(weights_.empty() ? synthetic_embed(t, dims_) : linear_embed(t, weights_, dims_));
        return out;
    }
    std::vector<std::vector<float>> embed_with_template(const std::vector<std::string> &texts, const std::string &tmpl) const {
        std::vector<std::string> t2;
        for (auto &t : texts) t2.push_back(tmpl + t); // embedding/mod.rs:126-143 prefix templating
        return embed(t2);
    }
    size_t dimensions() const { return dims_; }
    const std::vector<uint16_t> &weights() const { return weights_; } // synthetic-linear: W [LINEAR_H x dims] bf16

  private:
    std::string mode_;
    size_t dims_;
    std::vector<uint16_t> weights_;
};

// ---- src/index/recompute.rs ---------------------------------------------------------------------------
// Pruned index: brute force over re-embedded passages.  Texts are collected (with the early filter,
// :62-79) and embedded in batches of 100 (:86-93) on the host-side provider; the N dot products, the
// stable descending sort and take(k) (:96-109) run on the GPU (leann_scan_topk_device).
class RecomputeSearcher {
  public:
    static RecomputeSearcher load(const std::string &index_path, size_t dimensions) {
        RecomputeSearcher s;
        s.passages_ = PassageStore::open(index_path);
        s.id_map_ = read_id_map(index_path, s.passages_);
        s.dimensions_ = dimensions;
        return s;
    }
    std::vector<SearchResult> search(const std::vector<float> &query_embedding, const EmbeddingProvider &provider, size_t top_k,
                                     const MetadataFilter *filter) const {
        std::vector<std::string> texts;
        std::vector<size_t> valid;
        for (size_t idx = 0; idx < id_map_.size(); idx++) {
            try {
                Passage p = passages_.get(id_map_[idx]);
                if (filter && !filter->matches(p.metadata)) continue;
                texts.push_back(p.text);
                valid.push_back(idx);
            } catch (...) { continue; }
        }
        if (texts.empty()) return {};
        const size_t d = dimensions_, n = texts.size();
        std::vector<float> all(n * d);
        for (size_t b0 = 0; b0 < n; b0 += 100) { // batch_size = 100
            std::vector<std::string> batch(texts.begin() + b0, texts.begin() + std::min(n, b0 + 100));
            auto emb = provider.embed(batch);
            for (size_t i = 0; i < emb.size(); i++) std::copy(emb[i].begin(), emb[i].end(), all.begin() + (b0 + i) * d);
        }
        // scores + stable sort desc + take(k) on the device
        int ndev = 0;
        leann_device_count(&ndev);
        if (ndev < 1) throw Error("no HIP device visible. This library has no CPU fallback.");
        const size_t ld = (d + 3) & ~(size_t)3, k = std::min(top_k, n);
        void *dX = nullptr, *dQ = nullptr, *dK = nullptr, *dS = nullptr, *dC = nullptr;
        check(leann_device_malloc(0, n * ld * 4, &dX));
        check(leann_device_malloc(0, ld * 4, &dQ));
        check(leann_device_malloc(0, std::max<size_t>(k, 1) * 8, &dK));
        check(leann_device_malloc(0, std::max<size_t>(k, 1) * 4, &dS));
        check(leann_device_malloc(0, 4, &dC));
        std::vector<float> padded(n * ld, 0.0f), qp(ld, 0.0f);
        for (size_t i = 0; i < n; i++) std::copy(all.begin() + i * d, all.begin() + (i + 1) * d, padded.begin() + i * ld);
        std::copy(query_embedding.begin(), query_embedding.begin() + std::min(d, query_embedding.size()), qp.begin());
        check(leann_device_upload(dX, padded.data(), n * ld * 4));
        check(leann_device_upload(dQ, qp.data(), ld * 4));
        std::vector<uint64_t> keys(std::max<size_t>(k, 1));
        std::vector<float> scores(std::max<size_t>(k, 1));
        uint32_t cnt = 0;
        if (k > 0) {
            check(leann_scan_topk_device((const float *)dX, n, d, ld, (const float *)dQ, 1, k, nullptr, 0, (uint64_t *)dK, (float *)dS,
                                         (uint32_t *)dC, nullptr));
            check(leann_device_download(keys.data(), dK, k * 8));
            check(leann_device_download(scores.data(), dS, k * 4));
            check(leann_device_download(&cnt, dC, 4));
        }
        for (void *p : {dX, dQ, dK, dS, dC}) leann_device_free(p);
        std::vector<SearchResult> results;
        for (uint32_t i = 0; i < cnt; i++) {
            const std::string &id = id_map_[valid[keys[i]]];
            try {
                Passage p = passages_.get(id);
                results.push_back({id, scores[i], p.text, p.metadata}); // score = raw dot (recompute.rs:99)
            } catch (...) {}
        }
        return results;
    }
    size_t len() const { return id_map_.size(); }
    bool is_empty() const { return id_map_.empty(); }

  private:
    PassageStore passages_;
    std::vector<std::string> id_map_;
    size_t dimensions_ = 0;
};

} // namespace leann
