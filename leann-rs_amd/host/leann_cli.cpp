// leann_cli.cpp — `leann search` with the reference's flags and output formats (src/cli/search.rs:10-260),
// plus a minimal `leann build` over a passages JSONL so that an index directory can be produced offline
// (the reference's ingestion — file walking, chunking, HTTP embedding — is out of scope).
//
// Additive, because every real embedding provider needs the network (src/cli/search.rs:100-112):
//     --embedding-mode synthetic     deterministic hashed-token embedder (leann_host.hpp)
//     --query-vector-file FILE       raw little-endian f32[dims] query embedding
//     --device N                     HIP device ordinal
#include "leann_host.hpp"

#include <cstring>
#include <iostream>
#include <unistd.h>

using namespace leann;

struct SearchArgs {
    std::string query;
    std::optional<std::string> index;
    size_t top_k = 5, complexity = 64;
    bool show_metadata = false, hybrid = false, auto_hybrid = true, expand = true;
    std::optional<std::string> filter;
    float hybrid_alpha = 0.7f;
    bool compat_polarity = true; // additive: SURVEY.md N1 (see SearchOptions::compat_polarity)
    std::string format = "text";
    std::optional<std::string> query_prompt_template, embedding_mode, query_vector_file;
    std::string device = getenv("LEANN_DEVICES") ? getenv("LEANN_DEVICES") : "0"; // "0", or a list / range = sharded (leann_backend.h)
    bool device_filter = false;
};

static void usage_search() {
    puts("Query an index\n\nUsage: leann search [OPTIONS] <QUERY>\n\nArguments:\n  <QUERY>  Search query\n\nOptions:\n"
         "  -i, --index <INDEX>                Index name to search (defaults to current directory name)\n"
         "      --top-k <TOP_K>                Number of results to return [default: 5]\n"
         "      --complexity <COMPLEXITY>      Search complexity (higher = more accurate but slower) [default: 64]\n"
         "      --show-metadata                Show file paths in results\n"
         "  -f, --filter <FILTER>              Filter results by metadata (e.g., \"source:*.rs\" or \"type=code\")\n"
         "      --hybrid                       Enable hybrid search (vector + BM25)\n"
         "      --auto-hybrid <AUTO_HYBRID>    Auto-enable hybrid search for short queries (1-3 words) [default: true]\n"
         "      --expand <EXPAND>              Expand short queries with related terms for better recall [default: true]\n"
         "      --hybrid-alpha <HYBRID_ALPHA>  Weight for vector scores in hybrid mode (0.0-1.0, default 0.7) [default: 0.7]\n"
         "      --format <FORMAT>              Output format (text, json) [default: text] [possible values: text, json]\n"
         "      --embedding-api-key <KEY>      API key for embedding service [env: OPENAI_API_KEY]\n"
         "      --embedding-api-base <URL>     OpenAI API base URL [env: OPENAI_BASE_URL]\n"
         "      --embedding-host <HOST>        Ollama host for embeddings [env: OLLAMA_HOST]\n"
         "      --query-prompt-template <T>    Query prompt template prefix for asymmetric embedding models\n"
         "      --embedding-mode <MODE>        (additive) override the index's embedding mode; `synthetic` works offline\n"
         "      --query-vector-file <FILE>     (additive) raw f32 query embedding instead of embedding the query text\n"
         "      --device <SPEC>                (additive) HIP device ordinal, or a list / range (\"0-7\") = index sharded over several GPUs [env: LEANN_DEVICES] [default: 0]\n"
         "      --device-filter                (additive) evaluate --filter inside the GPU traversal instead of 5x over-fetch + post-filter\n"
         "  -h, --help                         Print help");
}

static bool parse_bool(const std::string &v) {
    if (v == "true") return true;
    if (v == "false") return false;
    throw Error("invalid value '" + v + "' for bool flag (expected true|false)");
}

static int run_search(int argc, char **argv) {
    SearchArgs a;
    bool have_query = false;
    for (int i = 0; i < argc; i++) {
        std::string s = argv[i];
        auto val = [&]() -> std::string {
            size_t eq = s.find('=');
            if (s.rfind("--", 0) == 0 && eq != std::string::npos) return s.substr(eq + 1);
            if (i + 1 >= argc) throw Error("a value is required for '" + s + "' but none was supplied");
            return argv[++i];
        };
        std::string name = s.rfind("--", 0) == 0 ? s.substr(0, s.find('=')) : s;
        if (name == "-h" || name == "--help") { usage_search(); return 0; }
        else if (name == "-i" || name == "--index") a.index = val();
        else if (name == "--top-k") a.top_k = std::stoul(val());
        else if (name == "--complexity") a.complexity = std::stoul(val());
        else if (name == "--show-metadata") a.show_metadata = true;
        else if (name == "--device-filter") a.device_filter = true;
        else if (name == "-f" || name == "--filter") a.filter = val();
        else if (name == "--hybrid") a.hybrid = true;
        else if (name == "--auto-hybrid") a.auto_hybrid = parse_bool(val());
        else if (name == "--expand") a.expand = parse_bool(val());
        else if (name == "--hybrid-alpha") a.hybrid_alpha = std::stof(val());
        else if (name == "--compat-polarity") a.compat_polarity = parse_bool(val());
        else if (name == "--format") { a.format = val(); if (a.format != "text" && a.format != "json") throw Error("invalid value '" + a.format + "' for '--format <FORMAT>' [possible values: text, json]"); }
        else if (name == "--embedding-api-key" || name == "--embedding-api-base" || name == "--embedding-host") (void)val();
        else if (name == "--query-prompt-template") a.query_prompt_template = val();
        else if (name == "--embedding-mode") a.embedding_mode = val();
        else if (name == "--query-vector-file") a.query_vector_file = val();
        else if (name == "--device") a.device = val();
        else if (name == "-v" || name == "--verbose" || name == "-q" || name == "--quiet") {}
        else if (!s.empty() && s[0] == '-' && s.size() > 1) throw Error("unexpected argument '" + s + "' found");
        else if (!have_query) { a.query = s; have_query = true; }
        else throw Error("unexpected argument '" + s + "' found");
    }
    if (!have_query) throw Error("the following required arguments were not provided:\n  <QUERY>");

    // search.rs:75-88
    std::string index_name;
    if (a.index) index_name = *a.index;
    else {
        char cwd[4096];
        std::string d = getcwd(cwd, sizeof cwd) ? cwd : "index";
        size_t sl = d.find_last_of('/');
        index_name = sl == std::string::npos ? d : d.substr(sl + 1);
        if (index_name.empty()) index_name = "index";
    }
    std::string index_dir = find_index(index_name);
    std::string meta_path = index_dir + "/documents.leann.meta.json", index_path = index_dir + "/documents.leann";
    IndexMeta meta = IndexMeta::load(meta_path);
    const bool is_pruned = meta.is_pruned;

    std::string mode = a.embedding_mode ? *a.embedding_mode : meta.embedding_mode;
    std::string query_template = a.query_prompt_template.value_or("");
    if (!a.query_prompt_template && !meta.embedding_options.is_null())
        if (auto *t = meta.embedding_options.get("query_prompt_template"))
            if (t->is_string()) query_template = t->s;

    std::optional<MetadataFilter> filter;
    if (a.filter) {
        filter = MetadataFilter::parse(*a.filter);
        if (!filter) throw Error("Invalid filter syntax: " + *a.filter);
    }
    size_t word_count = 0;
    { std::istringstream ws(a.query); for (std::string w; ws >> w;) word_count++; }
    const bool use_hybrid = a.hybrid || (a.auto_hybrid && word_count <= 3); // search.rs:147-148

    auto embed_query = [&](const std::string &text) -> std::vector<float> {
        if (a.query_vector_file) {
            std::string raw = read_file(*a.query_vector_file);
            if (raw.size() != meta.dimensions * 4) throw Error("--query-vector-file must hold " + std::to_string(meta.dimensions) + " f32 values");
            std::vector<float> v(meta.dimensions);
            memcpy(v.data(), raw.data(), raw.size());
            return v;
        }
        if (mode != "synthetic" && mode != "synthetic-linear") {
            if (mode == "openai" || mode == "ollama" || mode == "gemini")
                throw Error("Embedding mode '" + mode + "' needs a network service that is unavailable in this build; "
                            "pass --embedding-mode synthetic or --query-vector-file");
            throw Error("Unknown embedding mode in index: " + mode); // search.rs:112
        }
        return EmbeddingProvider(mode, meta.dimensions).embed_with_template({text}, query_template)[0];
    };

    std::vector<SearchResult> results;
    auto graph_search = [&]() {
        IndexSearcher searcher = IndexSearcher::load(index_path, meta, a.device.c_str());
        // query expansion (src/index/query.rs) is a text heuristic outside the scoped path: not applied
        auto q = embed_query(a.query);
        SearchOptions opts(a.top_k, a.complexity);
        if (filter) opts.with_filter(*filter);
        if (filter && a.device_filter) opts.with_device_filter(*a.filter);
        if (use_hybrid) opts.with_hybrid(a.query, a.hybrid_alpha);
        opts.with_compat_polarity(a.compat_polarity);
        results = searcher.search_with_options(q, opts);
    };
    // A pruned index (search.rs:151-167) recomputes embeddings at query time.  The reference scans every passage through the provider
    // (RecomputeSearcher).  When the directory holds a recompute-on GRAPH (`leann build --recompute-graph`: our file format version 2 —
    // graph + compact encoder inputs, no vectors) the graph is walked instead and distances are recomputed on the device per visited node.
    const std::string ann_file = with_extension(index_path, meta.backend_name == "diskann" ? "diskann" : "index");
    bool pruned_graph = false;
    if (is_pruned && file_exists(ann_file)) {
        std::ifstream f(ann_file, std::ios::binary);
        char hd[16] = {0};
        f.read(hd, 16);
        uint32_t version = 0;
        memcpy(&version, hd + 8, 4);
        pruned_graph = f.gcount() == 16 && !memcmp(hd, "LEANNGX1", 8) && version == 2;
    }
    if (is_pruned && !pruned_graph) {
        auto q = embed_query(a.query);
        if (mode != "synthetic" && mode != "synthetic-linear")
            throw Error("recompute mode needs an embedding provider; only `synthetic` / `synthetic-linear` work offline");
        EmbeddingProvider provider(mode, meta.dimensions);
        RecomputeSearcher searcher = RecomputeSearcher::load(index_path, meta.dimensions);
        results = searcher.search(q, provider, a.top_k, filter ? &*filter : nullptr);
    } else {
        graph_search();
    }

    if (a.format == "json") { // search.rs:211-223
        lj::Value arr = lj::Value::array();
        for (auto &r : results) {
            lj::Value o = lj::Value::object();
            o["id"] = lj::Value::string(r.id);
            o["score"] = lj::Value::number((double)r.score); // f32 widened, like serde_json's From<f32>
            o["text"] = lj::Value::string(r.text);
            o["metadata"] = r.metadata;
            arr.a->push_back(o);
        }
        printf("%s\n", lj::to_string_pretty(arr).c_str());
    } else { // search.rs:225-256
        printf("\nSearch results for '%s' (top %zu):\n\n", a.query.c_str(), results.size());
        for (size_t i = 0; i < results.size(); i++) {
            auto &r = results[i];
            printf("%zu. Score: %.4f\n", i + 1, (double)r.score);
            if (a.show_metadata) {
                if (auto *src = r.metadata.get("source")) printf("   Source: %s\n", lj::to_string(*src).c_str());
                if (r.metadata.is_object())
                    for (auto &kv : *r.metadata.o)
                        if (kv.first != "source") printf("   %s: %s\n", kv.first.c_str(), lj::to_string(kv.second).c_str());
            }
            std::string text = r.text;
            if (text.size() > 200) { // truncate on a UTF-8 boundary
                size_t end = 200;
                while (end > 0 && ((unsigned char)text[end] & 0xC0) == 0x80) end--;
                text = text.substr(0, end) + "...";
            }
            printf("   %s\n\n", text.c_str());
        }
    }
    return 0;
}

// leann build --index-dir DIR --passages-jsonl FILE [--backend-name hnsw|diskann] [--graph-degree 32]
//             [--complexity 64] [--dimensions 128] [--embedding-mode synthetic|synthetic-linear] [--recompute] [--pruned]
//             [--recompute-graph]
// --recompute       also write documents.embeddings (only in recompute mode, src/index/builder.rs:105-113), so that the index can be
//                   pruned later; off by default like the reference (meta.is_recompute = args.recompute)
// --pruned          no ANN file and no embeddings: the reference's pruned state (brute-force recompute at query time)
// --recompute-graph (needs --embedding-mode synthetic-linear) graph + compact encoder inputs, no vectors
static int run_build(int argc, char **argv) {
    std::string dir, jsonl, backend_name = "hnsw", mode = "synthetic";
    size_t degree = 32, complexity = 64, dims = 128;
    bool pruned = false, recompute = false, rgraph = false; // is_recompute = args.recompute, default false (src/cli/build.rs:363)
    for (int i = 0; i < argc; i++) {
        std::string s = argv[i];
        auto val = [&]() -> std::string { if (i + 1 >= argc) throw Error("missing value for " + s); return argv[++i]; };
        if (s == "--index-dir") dir = val();
        else if (s == "--passages-jsonl" || s == "--docs") jsonl = val();
        else if (s == "--backend-name") backend_name = val();
        else if (s == "--graph-degree") degree = std::stoul(val());
        else if (s == "--complexity") complexity = std::stoul(val());
        else if (s == "--dimensions") dims = std::stoul(val());
        else if (s == "--embedding-mode") mode = val();
        else if (s == "--pruned") pruned = true;
        else if (s == "--recompute") recompute = true;
        else if (s == "--recompute-graph") { rgraph = true; mode = "synthetic-linear"; }
        else throw Error("unexpected argument '" + s + "' found");
    }
    if (dir.empty() || jsonl.empty()) throw Error("usage: leann build --index-dir DIR --passages-jsonl FILE [...]");
    int backend = backend_name == "hnsw" ? LEANN_BACKEND_HNSW : backend_name == "diskann" ? LEANN_BACKEND_DISKANN : -1;
    if (backend < 0) throw Error("Unknown backend: " + backend_name);
    EmbeddingProvider provider(mode, dims);
    ::mkdir(dir.c_str(), 0755);
    std::string stem = dir + "/documents.leann";
    std::ifstream in(jsonl);
    if (!in) throw Error("cannot open " + jsonl);
    PassageStoreWriter w(stem);
    std::vector<float> all;
    std::vector<uint16_t> feats;
    std::ofstream ids(with_extension(stem, "ids.txt"));
    size_t n = 0;
    for (std::string line; std::getline(in, line);) {
        if (line.empty()) continue;
        lj::Value v = lj::parse(line);
        Passage p;
        p.id = v.get("id") ? v.get("id")->s : std::to_string(n + 1); // chunk ids start at "1" (chunker/simple.rs:38-40)
        p.text = v.get("text") ? v.get("text")->s : "";
        p.metadata = v.get("metadata") ? *v.get("metadata") : lj::Value::object();
        w.add(p);
        ids << p.id << "\n";
        if (rgraph) {
            auto f = linear_features(p.text);
            feats.insert(feats.end(), f.begin(), f.end());
        } else if (!pruned) {
            auto e = provider.embed({p.text})[0];
            all.insert(all.end(), e.begin(), e.end());
        }
        n++;
    }
    w.finish();
    ids.close();
    if (rgraph) {
        // the embeddings exist only transiently on the device while the graph is built (leann_recompute_build_index)
        leann_recompute *r = nullptr;
        check(leann_recompute_create_host(feats.data(), n, LINEAR_H, provider.weights().data(), dims, 0, 0, &r));
        leann_backend *h = nullptr;
        int rc = leann_recompute_build_index(r, backend, degree, complexity, &h);
        if (rc == 0) rc = leann_backend_save(h, stem.c_str());
        if (h) leann_backend_close(h);
        leann_recompute_close(r);
        check(rc);
    } else if (!pruned) {
        if (recompute) { // src/index/embeddings.rs: raw LE f32 [n x dims]
            std::ofstream ef(with_extension(stem, "embeddings"), std::ios::binary);
            ef.write((const char *)all.data(), (std::streamsize)(all.size() * 4));
        }
        check(leann_backend_build(backend, all.data(), n, dims, degree, complexity, stem.c_str()));
    }
    IndexMeta m;
    m.version = "1.0";
    m.backend_name = backend_name;
    m.embedding_model = mode == "synthetic" ? "synthetic-hash" : "synthetic-linear-256";
    m.embedding_mode = mode;
    m.dimensions = dims;
    m.passage_count = n;
    m.is_recompute = pruned || rgraph || recompute;
    m.is_pruned = pruned || rgraph;
    if (rgraph) {
        m.backend_kwargs = lj::Value::object();
        m.backend_kwargs["recompute_graph"] = lj::Value::boolean(true);
        m.backend_kwargs["feature_dim"] = lj::Value::integer((int64_t)LINEAR_H);
    }
    m.save(dir + "/documents.leann.meta.json");
    printf("Indexed %zu passages (%zu dims, backend %s%s) into %s\n", n, dims, backend_name.c_str(),
           rgraph ? ", recompute-on graph (no vectors)" : pruned ? ", pruned" : "", dir.c_str());
    return 0;
}

int main(int argc, char **argv) {
    try {
        if (argc < 2 || !strcmp(argv[1], "-h") || !strcmp(argv[1], "--help")) {
            puts("LEANN search path on MI355X\n\nUsage: leann <COMMAND>\n\nCommands:\n  search  Query an index\n  build   Build an index from a passages JSONL (synthetic embeddings)\n");
            return argc < 2 ? 2 : 0;
        }
        if (!strcmp(argv[1], "--version") || !strcmp(argv[1], "-V")) { printf("leann %s\n", leann_version()); return 0; }
        if (!strcmp(argv[1], "search")) return run_search(argc - 2, argv + 2);
        if (!strcmp(argv[1], "build")) return run_build(argc - 2, argv + 2);
        throw Error(std::string("unrecognized subcommand '") + argv[1] + "'");
    } catch (const std::exception &e) {
        fprintf(stderr, "Error: %s\n", e.what());
        return 1;
    }
}
