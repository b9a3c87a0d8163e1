// host_selftest.cpp — dumps the host-side (CPU) results of the C++ index-layer mirror as JSON so that
// pytest can compare them with tests/golden and the oracle.  No GPU calls.
//   host_selftest <cases.json>   ->  stdout JSON
#include "leann_host.hpp"
#include <iostream>
using namespace leann;

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: host_selftest cases.json\n"); return 2; }
    lj::Value in = lj::parse(read_file(argv[1]));
    lj::Value out = lj::Value::object();
    // tokenize
    lj::Value tok = lj::Value::object();
    for (auto &kv : *in["tokenize"].o) {
        lj::Value a = lj::Value::array();
        for (auto &t : tokenize(kv.first)) a.a->push_back(lj::Value::string(t));
        tok[kv.first] = a;
    }
    out["tokenize"] = tok;
    // bm25
    lj::Value bm = lj::Value::object();
    for (auto &kv : *in["bm25"].o) {
        std::vector<std::string> docs;
        for (auto &d : *kv.second.get("docs")->a) docs.push_back(d.s);
        Bm25Scorer s = Bm25Scorer::build(docs);
        lj::Value r = lj::Value::object(), sc = lj::Value::array(), top = lj::Value::array();
        for (float f : s.score_query(kv.second.get("query")->s)) sc.a->push_back(lj::Value::number((double)f));
        for (auto &p : s.search(kv.second.get("query")->s, 2)) top.a->push_back(lj::Value::integer((int64_t)p.first));
        r["scores"] = sc;
        r["top2"] = top;
        bm[kv.first] = r;
    }
    out["bm25"] = bm;
    // hybrid_rerank
    lj::Value hr = lj::Value::array();
    for (auto &c : *in["hybrid_rerank"].a) {
        std::vector<std::pair<size_t, float>> vr;
        for (auto &p : *c.get("vr")->a) vr.emplace_back((size_t)(*p.a)[0].as_f64(), (float)(*p.a)[1].as_f64());
        std::vector<float> b;
        for (auto &x : *c.get("bm")->a) b.push_back((float)x.as_f64());
        lj::Value o = lj::Value::array();
        for (auto &p : hybrid_rerank(vr, b, (float)c.get("alpha")->as_f64())) {
            lj::Value e = lj::Value::array();
            e.a->push_back(lj::Value::integer((int64_t)p.first));
            e.a->push_back(lj::Value::number((double)p.second));
            o.a->push_back(e);
        }
        hr.a->push_back(o);
    }
    out["hybrid_rerank"] = hr;
    // filters: [{filter, metadata, expect}]
    if (in.get("filters")) {
        lj::Value fr = lj::Value::array();
        for (auto &c : *in["filters"].a) {
            auto f = MetadataFilter::parse(c.get("filter")->s);
            fr.a->push_back(f ? lj::Value::boolean(f->matches(*c.get("metadata"))) : lj::Value());
        }
        out["filters"] = fr;
    }
    if (in.get("synthetic_embed")) {
        lj::Value e = lj::Value::array();
        for (float f : synthetic_embed(in["synthetic_embed"].s, 16)) e.a->push_back(lj::Value::number((double)f));
        out["synthetic_embed"] = e;
    }
    // IndexSearcher::search_with_options after the backend call, on recorded backend output:
    //   {"index_path": ".../documents.leann", "cases": [{top_k, hybrid, alpha, filter, query_text, backend: [[key, dist], ...]}]}
    if (in.get("searcher")) {
        const lj::Value &sv = in["searcher"];
        IndexSearcher searcher = IndexSearcher::load_passages_only(sv.get("index_path")->s);
        lj::Value all = lj::Value::array();
        for (auto &c : *sv.get("cases")->a) {
            const size_t top_k = (size_t)c.get("top_k")->as_f64();
            SearchOptions opts(top_k, 64);
            const bool hybrid = c.get("hybrid") && c.get("hybrid")->b;
            if (hybrid) opts.with_hybrid(c.get("query_text")->s, (float)c.get("alpha")->as_f64());
            if (c.get("compat_polarity")) opts.with_compat_polarity(c.get("compat_polarity")->b);
            if (c.get("filter") && c.get("filter")->is_string()) {
                auto f = MetadataFilter::parse(c.get("filter")->s);
                if (f) opts.with_filter(*f);
            }
            const size_t fetch_k = (opts.filter || opts.hybrid) ? top_k * 5 : top_k;
            std::vector<std::pair<size_t, float>> vr;
            for (auto &p : *c.get("backend")->a) vr.emplace_back((size_t)(*p.a)[0].as_f64(), (float)(*p.a)[1].as_f64());
            lj::Value res = lj::Value::array();
            for (auto &r : searcher.assemble_results(vr, opts, fetch_k)) {
                lj::Value e = lj::Value::array();
                e.a->push_back(lj::Value::string(r.id));
                e.a->push_back(lj::Value::number((double)r.score));
                res.a->push_back(e);
            }
            all.a->push_back(res);
        }
        out["searcher"] = all;
    }
    printf("%s\n", lj::to_string_pretty(out).c_str());
    return 0;
}
