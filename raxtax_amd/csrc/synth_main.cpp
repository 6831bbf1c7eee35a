// raxtax-synth: the synthetic inputs of SURVEY.md section 8d as FASTA files, without Python -- the C++ twin of
// raxtax_amd/synth.py (same "phylo" model, same parameters; its own PRNG, so the sequences are not those of the numpy
// generator: what the two share is the model, checked by tests/test_synth_cli.py on composition, divergence per level,
// taxonomy shape and the shares of exact copies and of queries with N).
//
//   raxtax-synth db <n_refs> <out.fasta> [--length 658] [--seed-root 1] [--seed-db 2]
//   raxtax-synth queries <db.fasta> <n_queries> <out.fasta> [--seed 3] [--mu 0.02] [--exact 0.10] [--n-frac 0.01]
//
// Model: a root sequence i.i.d. from p(A,C,G,T) = (0.263, 0.169, 0.143, 0.425) evolves down seven levels (phylum, class,
// order, family, genus, species, individual) with per-site substitution probabilities (0.06, 0.05, 0.04, 0.03, 0.03, 0.02,
// 0.005); a substituted site is redrawn from p.  Headers `>r{i};tax=p:P..,c:C..,o:O..,f:F..,g:G..,s:S..;` (what
// src/parser.rs:38-40 expects).  Queries: a uniformly chosen reference, mu_q per site; a share of exact copies; a share with
// one to three N.  PRNG: xoshiro256** seeded through splitmix64.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

struct Rng {  // xoshiro256** (Blackman & Vigna), state from splitmix64(seed)
    uint64_t s[4];
    explicit Rng(uint64_t seed) {
        for (auto &w : s) {
            seed += 0x9E3779B97F4A7C15ull;
            uint64_t z = seed;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            w = z ^ (z >> 31);
        }
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }  // [0, 1)
    uint64_t below(uint64_t n) { return (uint64_t)(uniform() * (double)n); }            // n < 2^53
};

const double kP[4] = {0.263, 0.169, 0.143, 0.425};
const double kMu[7] = {0.06, 0.05, 0.04, 0.03, 0.03, 0.02, 0.005};
const char *kPrefix[6] = {"p:P", "c:C", "o:O", "f:F", "g:G", "s:S"};
const char kLetters[5] = "ACGT";

uint8_t draw(Rng &r) {
    const double u = r.uniform();
    return (uint8_t)((u >= kP[0]) + (u >= kP[0] + kP[1]) + (u >= kP[0] + kP[1] + kP[2]));
}

void mutate(Rng &r, const uint8_t *src, uint8_t *dst, size_t len, double mu) {
    for (size_t i = 0; i < len; i++) dst[i] = r.uniform() < mu ? draw(r) : src[i];
}

void fanouts_for(uint64_t n, int (&fan)[6]) {  // synth.default_fanouts
    static const int a[6] = {2, 2, 2, 2, 2, 2}, b[6] = {3, 3, 3, 4, 4, 3}, c[6] = {3, 4, 5, 6, 6, 6}, d[6] = {4, 5, 6, 9, 10, 12};
    const int *f = n <= 2000 ? a : (n <= 100000 ? b : (n <= 1000000 ? c : d));
    for (int i = 0; i < 6; i++) fan[i] = f[i];
}

const char *opt(int argc, char **argv, const char *name, const char *dflt) {
    for (int i = 0; i + 1 < argc; i++)
        if (!strcmp(argv[i], name)) return argv[i + 1];
    return dflt;
}

int make_db(int argc, char **argv) {
    if (argc < 4) return 2;
    const uint64_t n = strtoull(argv[2], nullptr, 10);
    const size_t len = strtoull(opt(argc, argv, "--length", "658"), nullptr, 10);
    if (n == 0 || len == 0) return 2;
    Rng r_root(strtoull(opt(argc, argv, "--seed-root", "1"), nullptr, 10)), r(strtoull(opt(argc, argv, "--seed-db", "2"), nullptr, 10));
    int fan[6];
    fanouts_for(n, fan);
    std::vector<uint8_t> level(len), next;
    for (auto &b : level) b = draw(r_root);
    std::vector<std::string> labels{""}, next_labels;
    for (int d = 0; d < 6; d++) {
        const size_t parents = labels.size();
        next.resize(parents * fan[d] * len);
        next_labels.clear();
        size_t k = 0;
        for (size_t p = 0; p < parents; p++)
            for (int c = 0; c < fan[d]; c++, k++) {
                mutate(r, level.data() + p * len, next.data() + k * len, len, kMu[d]);
                next_labels.push_back((labels[p].empty() ? std::string() : labels[p] + ",") + kPrefix[d] + std::to_string(k));
            }
        level.swap(next);
        labels.swap(next_labels);
    }
    const uint64_t n_species = labels.size();
    FILE *f = fopen(argv[3], "w");
    if (!f) { perror(argv[3]); return 1; }
    std::vector<uint8_t> ind(len);
    std::string line(len, 'A');
    uint64_t idx = 0;
    for (uint64_t s = 0; s < n_species; s++) {
        const uint64_t per = n / n_species + (s < n % n_species ? 1 : 0);  // the first species take the remainder, as synth.make_db
        for (uint64_t i = 0; i < per; i++, idx++) {
            mutate(r, level.data() + s * len, ind.data(), len, kMu[6]);
            for (size_t j = 0; j < len; j++) line[j] = kLetters[ind[j]];
            fprintf(f, ">r%llu;tax=%s;\n%s\n", (unsigned long long)idx, labels[s].c_str(), line.c_str());
        }
    }
    if (fclose(f)) { perror(argv[3]); return 1; }
    fprintf(stderr, "raxtax-synth: %llu references, %llu species, length %zu -> %s\n", (unsigned long long)idx, (unsigned long long)n_species, len, argv[3]);
    return 0;
}

int make_queries(int argc, char **argv) {
    if (argc < 5) return 2;
    FILE *in = fopen(argv[2], "r");
    if (!in) { perror(argv[2]); return 1; }
    std::vector<std::string> seqs;
    {
        std::string cur;
        char buf[1 << 16];
        bool have = false;
        while (fgets(buf, sizeof buf, in)) {
            size_t l = strlen(buf);
            while (l && (buf[l - 1] == '\n' || buf[l - 1] == '\r')) buf[--l] = 0;
            if (buf[0] == '>') {
                if (have) seqs.push_back(cur);
                cur.clear();
                have = true;
            } else {
                cur += buf;
            }
        }
        if (have) seqs.push_back(cur);
        fclose(in);
    }
    if (seqs.empty()) { fprintf(stderr, "raxtax-synth: no records in %s\n", argv[2]); return 1; }
    const uint64_t nq = strtoull(argv[3], nullptr, 10);
    Rng r(strtoull(opt(argc, argv, "--seed", "3"), nullptr, 10));
    const double mu = atof(opt(argc, argv, "--mu", "0.02")), exact = atof(opt(argc, argv, "--exact", "0.10")), nfrac = atof(opt(argc, argv, "--n-frac", "0.01"));
    FILE *f = fopen(argv[4], "w");
    if (!f) { perror(argv[4]); return 1; }
    for (uint64_t q = 0; q < nq; q++) {
        std::string s = seqs[r.below(seqs.size())];
        if (!(r.uniform() < exact))
            for (auto &ch : s)
                if (r.uniform() < mu) ch = kLetters[draw(r)];
        if (r.uniform() < nfrac) {
            const uint64_t k = 1 + r.below(3);
            for (uint64_t i = 0; i < k; i++) s[r.below(s.size())] = 'N';
        }
        fprintf(f, ">q%llu\n%s\n", (unsigned long long)q, s.c_str());
    }
    if (fclose(f)) { perror(argv[4]); return 1; }
    fprintf(stderr, "raxtax-synth: %llu queries from %zu references -> %s\n", (unsigned long long)nq, seqs.size(), argv[4]);
    return 0;
}

}  // namespace

int main(int argc, char **argv) {
    int rc = 2;
    if (argc >= 2 && !strcmp(argv[1], "db")) rc = make_db(argc, argv);
    else if (argc >= 2 && !strcmp(argv[1], "queries")) rc = make_queries(argc, argv);
    if (rc == 2)
        fprintf(stderr, "usage: raxtax-synth db <n_refs> <out.fasta> [--length 658] [--seed-root 1] [--seed-db 2]\n"
                        "       raxtax-synth queries <db.fasta> <n_queries> <out.fasta> [--seed 3] [--mu 0.02] [--exact 0.10] [--n-frac 0.01]\n");
    return rc;
}
