// mtsv-build -- drop-in command line of src/bin/mtsv-build.rs over libmtsv_amd: FASTA database ->
// MG-index file (bincode layout of MGIndex, src/index.rs:60-68).  Same flags and defaults
// (-f/--fasta, -i/--index, --sa-sample 32, --sample-interval 64, --mapping, --skip-missing, -v),
// same exit codes (0 ok, 1 error; invalid numbers abort like the reference's panics, 101).
// Extra: --device N builds the suffix array on that GPU (default: device 0 when one is visible,
// --device -1 forces the host threads), --threads N.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/mtsv_amd.h"

namespace {
bool g_verbose = false;
void logmsg(const char* level, const std::string& msg) {
    if (!g_verbose && !strcmp(level, "DEBUG")) return;
    char ts[32];
    time_t t = time(nullptr);
    strftime(ts, sizeof ts, "%Y-%m-%d %H:%M:%S", localtime(&t));
    printf("[%s %s mtsv_build] %s\n", level, ts, msg.c_str());
    fflush(stdout);
}
[[noreturn]] void panic(const std::string& m) {
    fprintf(stderr, "thread 'main' panicked: %s\n", m.c_str());
    exit(101);
}
bool parse_u32(const std::string& s, uint32_t* out) {
    if (s.empty()) return false;
    uint64_t v = 0;
    for (char c : s) {
        if (c < '0' || c > '9') return false;
        v = v * 10 + (uint64_t)(c - '0');
        if (v > 0xffffffffull) return false;
    }
    *out = (uint32_t)v;
    return true;
}
std::string trim(const std::string& s) {
    size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? "" : s.substr(a, b - a + 1);
}
// split_mapping_line, src/io.rs:28-33
std::vector<std::string> split(const std::string& line, char delim) {
    std::vector<std::string> out;
    if (delim) {
        size_t p = 0;
        for (;;) {
            size_t d = line.find(delim, p);
            out.push_back(trim(line.substr(p, d == std::string::npos ? std::string::npos : d - p)));
            if (d == std::string::npos) break;
            p = d + 1;
        }
    } else {
        size_t p = 0;
        while (p < line.size()) {
            while (p < line.size() && isspace((unsigned char)line[p])) p++;
            size_t e = p;
            while (e < line.size() && !isspace((unsigned char)line[e])) e++;
            if (e > p) out.push_back(line.substr(p, e - p));
            p = e;
        }
    }
    return out;
}
}  // namespace

int main(int argc, char** argv) {
    std::string fasta, index, sa = "32", fm = "64", mapping;
    bool skip_missing = false;
    int device = -2, threads = 16;
    for (int i = 1; i < argc; i++) {
        std::string k = argv[i];
        auto val = [&]() -> std::string {
            if (i + 1 >= argc) {
                fprintf(stderr, "error: The argument '%s' requires a value but none was supplied\n", k.c_str());
                exit(1);
            }
            return argv[++i];
        };
        if (k == "-f" || k == "--fasta") fasta = val();
        else if (k == "-i" || k == "--index") index = val();
        else if (k == "--sa-sample") sa = val();
        else if (k == "--sample-interval") fm = val();
        else if (k == "--mapping") mapping = val();
        else if (k == "--skip-missing") skip_missing = true;
        else if (k == "-v") g_verbose = true;
        else if (k == "--device") device = atoi(val().c_str());
        else if (k == "--threads") threads = atoi(val().c_str());
        else if (k == "-h" || k == "--help") {
            printf("mtsv-build -f <FASTA> -i <INDEX> [--sa-sample 32] [--sample-interval 64] [--mapping FILE] [--skip-missing] [-v] [--device N] [--threads N]\n");
            return 0;
        } else {
            fprintf(stderr, "error: Found argument '%s' which wasn't expected, or isn't valid in this context\n", k.c_str());
            return 1;
        }
    }
    if (fasta.empty() || index.empty()) {
        fprintf(stderr, "error: The following required arguments were not provided: --fasta <FASTA> --index <INDEX>\n");
        return 1;
    }
    uint32_t fm_k = 0, sa_s = 0;
    if (!parse_u32(fm, &fm_k)) panic("Invalid index sample interval entered!");
    if (!parse_u32(sa, &sa_s)) panic("Invalid suffix array sample interval entered!");
    if (skip_missing && mapping.empty()) logmsg("WARN", "--skip-missing has no effect without --mapping.");
    if (device == -2) device = mtsv_device_count() > 0 ? 0 : -1;
    mtsv_set_build_device(device);

    mtsv_index* ix = nullptr;
    int rc;
    if (mapping.empty()) {
        logmsg("DEBUG", "Opening FASTA database file...");
        rc = mtsv_index_build_fasta(fasta.c_str(), fm_k, sa_s, threads, &ix);
    } else {
        // parse_header_mapping (src/io.rs:36-112) + parse_fasta_db_with_mapping (:153-185)
        std::ifstream mf(mapping);
        if (!mf) {
            logmsg("ERROR", "Error parsing mapping file: cannot open " + mapping);
            return 1;
        }
        std::string line, header_line;
        while (std::getline(mf, line))
            if (!trim(line).empty()) {
                header_line = line;
                break;
            }
        if (header_line.empty()) {
            logmsg("ERROR", "Error parsing mapping file: Empty mapping file");
            return 1;
        }
        char delim = 0;
        for (char c : {',', '\t', ';', '|'})
            if (header_line.find(c) != std::string::npos) {
                delim = c;
                break;
            }
        auto cols = split(header_line, delim);
        int hi = -1, ti = -1, si = -1;
        for (size_t c = 0; c < cols.size(); c++) {
            std::string f = trim(cols[c]);
            for (auto& ch : f) ch = (char)tolower((unsigned char)ch);
            if (f == "header" && hi < 0) hi = (int)c;
            if (f == "taxid" && ti < 0) ti = (int)c;
            if ((f == "seqid" || f == "gi") && si < 0) si = (int)c;
        }
        if (hi < 0 || ti < 0 || si < 0) {
            logmsg("ERROR", "Error parsing mapping file: Missing 'header', 'taxid' or 'seqid' column in mapping file");
            return 1;
        }
        std::unordered_map<std::string, std::pair<uint32_t, uint32_t>> map;  // header -> (gi, taxid)
        while (std::getline(mf, line)) {
            std::string t = trim(line);
            if (t.empty()) continue;
            auto f = split(t, delim);
            size_t need = (size_t)std::max(hi, std::max(ti, si));
            uint32_t tax, gi;
            if (f.size() <= need || trim(f[hi]).empty() || !parse_u32(f[ti], &tax) || !parse_u32(f[si], &gi) || map.count(trim(f[hi]))) {
                logmsg("ERROR", "Error parsing mapping file: invalid or duplicate row: " + t);
                return 1;
            }
            map[trim(f[hi])] = {gi, tax};
        }
        std::ifstream in(fasta, std::ios::binary);
        if (!in) panic("Unable to open FASTA database for parsing.");
        std::vector<std::string> seqs;
        std::vector<uint32_t> taxs, gis;
        bool keep = false;
        while (std::getline(in, line)) {
            if (!line.empty() && line.back() == '\r') line.pop_back();
            if (line.empty()) continue;
            if (line[0] == '>') {
                size_t e = line.find_first_of(" \t", 1);
                std::string id = line.substr(1, e == std::string::npos ? std::string::npos : e - 1);
                auto it = map.find(id);
                if (it == map.end()) {
                    if (!skip_missing) {
                        logmsg("ERROR", "Error building index: Missing mapping for header " + id);
                        return 1;
                    }
                    logmsg("WARN", "Missing mapping for header " + id + ", skipping.");
                    keep = false;
                    continue;
                }
                keep = true;
                seqs.emplace_back();
                gis.push_back(it->second.first);
                taxs.push_back(it->second.second);
            } else if (keep) {
                seqs.back() += line;
            }
        }
        std::vector<const uint8_t*> ptrs(seqs.size());
        std::vector<uint64_t> lens(seqs.size());
        for (size_t k = 0; k < seqs.size(); k++) {
            ptrs[k] = (const uint8_t*)seqs[k].data();
            lens[k] = seqs[k].size();
        }
        rc = mtsv_index_build(seqs.size(), taxs.data(), gis.data(), ptrs.data(), lens.data(), fm_k, sa_s, threads, &ix);
    }
    if (rc == MTSV_E_IO && std::string(mtsv_last_error()).find("cannot open") != std::string::npos)
        panic("Unable to open FASTA database for parsing.");
    if (rc != MTSV_OK) {
        logmsg("ERROR", std::string("Error building index: ") + mtsv_last_error());
        return 1;
    }
    logmsg("INFO", "File parsed, building index...");
    logmsg("INFO", "Writing index to file...");
    if (mtsv_index_write(ix, index.c_str()) != MTSV_OK) {
        logmsg("ERROR", std::string("Error building index: ") + mtsv_last_error());
        return 1;
    }
    mtsv_index_free(ix);
    logmsg("INFO", "Done building and writing index!");
    return 0;
}
