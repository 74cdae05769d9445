#include "fileReadingFunctions.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>

namespace {
// positions the stream just after the first token equal to `tag`
bool seek_tag(FILE *f, const char *tag)
{
    char tok[256];
    while (std::fscanf(f, "%255s", tok) == 1)
        if (std::strcmp(tok, tag) == 0) return true;
    return false;
}
}  // namespace

// fileReadingFunctions.cpp:6-32 -- "$Nodes", count, then (id x y z) per node
std::vector<std::tuple<double, double, double>> pointsFromMshFile(const char *fname)
{
    std::vector<std::tuple<double, double, double>> pts;
    FILE *f = std::fopen(fname, "r");
    if (!f) return pts;
    int nv = 0;
    if (seek_tag(f, "$Nodes") && std::fscanf(f, "%d", &nv) == 1) {
        pts.reserve((size_t)nv);
        for (int i = 0; i < nv; ++i) {
            int id;
            double x, y, z;
            if (std::fscanf(f, "%d %lf %lf %lf", &id, &x, &y, &z) != 4) break;
            pts.emplace_back(x, y, z);
        }
    }
    std::fclose(f);
    return pts;
}

// fileReadingFunctions.cpp:33-57 -- same without the node id column
std::vector<std::tuple<double, double, double>> pointsFromTxts(const char *fname)
{
    std::vector<std::tuple<double, double, double>> pts;
    FILE *f = std::fopen(fname, "r");
    if (!f) return pts;
    int nv = 0;
    if (seek_tag(f, "$Nodes") && std::fscanf(f, "%d", &nv) == 1) {
        pts.reserve((size_t)nv);
        for (int i = 0; i < nv; ++i) {
            double x, y, z;
            if (std::fscanf(f, "%lf %lf %lf", &x, &y, &z) != 3) break;
            pts.emplace_back(x, y, z);
        }
    }
    std::fclose(f);
    return pts;
}

// fileReadingFunctions.cpp:58-69 -- reads nv integers and returns an EMPTY vector
// (the reference never stores them); kept for surface compatibility.
std::vector<int> orderFromTxt(const char *fname, int nv)
{
    std::vector<int> order;
    FILE *f = std::fopen(fname, "r");
    if (!f) return order;
    int v;
    for (int i = 0; i < nv; ++i)
        if (std::fscanf(f, "%d", &v) != 1) break;
    std::fclose(f);
    return order;
}

// fileReadingFunctions.cpp:70-79 -- one value per line, default ostream precision
void writeVectorToTxt(std::vector<double> vec, const char *filename)
{
    std::ofstream out(filename);
    for (double v : vec) out << v << "\n";
}

// fileReadingFunctions.cpp:80-150 -- for every boundary point the (up to two)
// other boundary points it shares a triangle with.  Element lines are
// "id type ntags tags... nodes..."; the reference assumes 2 tags.
std::vector<std::pair<int, int>> boundPtsConnFromMsh(const char *fname, const std::vector<int> &bcFlags)
{
    std::vector<std::pair<int, int>> conn(bcFlags.size(), std::make_pair(-1, -1));
    FILE *f = std::fopen(fname, "r");
    if (!f) return conn;
    int ne = 0;
    if (seek_tag(f, "$Elements") && std::fscanf(f, "%d", &ne) == 1) {
        for (int e = 0; e < ne; ++e) {
            int id, type, t;
            if (std::fscanf(f, "%d %d", &id, &type) != 2) break;
            if (type == 2) {
                for (int k = 0; k < 3; ++k) if (std::fscanf(f, "%d", &t) != 1) t = 0;  // ntags + 2 tags
                std::vector<int> bnd;
                for (int k = 0; k < 3; ++k) {
                    int node = 0;
                    if (std::fscanf(f, "%d", &node) != 1) break;
                    --node;
                    if (bcFlags.at((size_t)node) != 0) bnd.push_back(node);
                }
                if (bnd.size() >= 2) {
                    for (int me : bnd)
                        for (int other : bnd) {
                            if (other == me) continue;
                            if (conn[(size_t)me].first == -1) conn[(size_t)me].first = other;
                            else if (conn[(size_t)me].second == -1) conn[(size_t)me].second = other;
                        }
                }
            } else if (type == 1) {
                for (int k = 0; k < 5; ++k) if (std::fscanf(f, "%d", &t) != 1) break;
            } else if (type == 15) {
                for (int k = 0; k < 4; ++k) if (std::fscanf(f, "%d", &t) != 1) break;
            }
        }
    }
    std::fclose(f);
    return conn;
}

bool writePointsToMshFile(const char *fname, const std::vector<std::tuple<double, double, double>> &pts,
                          const std::vector<int> *tri)
{
    FILE *f = std::fopen(fname, "w");
    if (!f) return false;
    std::fprintf(f, "$MeshFormat\n2.2 0 8\n$EndMeshFormat\n$Nodes\n%zu\n", pts.size());
    for (size_t i = 0; i < pts.size(); ++i)
        std::fprintf(f, "%zu %.17g %.17g %.17g\n", i + 1, std::get<0>(pts[i]), std::get<1>(pts[i]), std::get<2>(pts[i]));
    std::fprintf(f, "$EndNodes\n");
    if (tri) {
        const size_t nt = tri->size() / 3;
        std::fprintf(f, "$Elements\n%zu\n", nt);
        for (size_t k = 0; k < nt; ++k)
            std::fprintf(f, "%zu 2 2 0 1 %d %d %d\n", k + 1, (*tri)[3 * k], (*tri)[3 * k + 1], (*tri)[3 * k + 2]);
        std::fprintf(f, "$EndElements\n");
    }
    std::fclose(f);
    return true;
}

// ---- binary cloud container (not in the reference; fileReadingFunctions.h) ------------------
namespace {
struct BinHeader {
    char magic[8];
    unsigned int version, dim;
    unsigned long long n;
};
static_assert(sizeof(BinHeader) == 24, "binary cloud header layout");
const char kBinMagic[8] = {'M', 'M', 'G', 'C', 'L', 'O', 'U', 'D'};
}  // namespace

bool writePointsToBinFile(const char *fname, const std::vector<std::tuple<double, double, double>> &pts, int dim)
{
    FILE *f = std::fopen(fname, "wb");
    if (!f) return false;
    BinHeader h;
    std::memcpy(h.magic, kBinMagic, 8);
    h.version = 1;
    h.dim = (unsigned)dim;
    h.n = pts.size();
    bool ok = std::fwrite(&h, sizeof(h), 1, f) == 1;
    std::vector<double> buf;
    const size_t chunk = 1 << 20;
    for (size_t b = 0; ok && b < pts.size(); b += chunk) {
        const size_t e = std::min(pts.size(), b + chunk);
        buf.resize((e - b) * 3);
        for (size_t i = b; i < e; ++i) {
            buf[(i - b) * 3] = std::get<0>(pts[i]);
            buf[(i - b) * 3 + 1] = std::get<1>(pts[i]);
            buf[(i - b) * 3 + 2] = std::get<2>(pts[i]);
        }
        ok = std::fwrite(buf.data(), sizeof(double), buf.size(), f) == buf.size();
    }
    return std::fclose(f) == 0 && ok;
}

std::vector<std::tuple<double, double, double>> pointsFromBinFile(const char *fname, int *dim)
{
    std::vector<std::tuple<double, double, double>> pts;
    FILE *f = std::fopen(fname, "rb");
    if (!f) return pts;
    BinHeader h;
    if (std::fread(&h, sizeof(h), 1, f) != 1 || std::memcmp(h.magic, kBinMagic, 8) != 0 || h.version != 1 ||
        (h.dim != 2 && h.dim != 3)) {
        std::fclose(f);
        return pts;
    }
    // the count is checked against the file size before anything is allocated
    std::fseek(f, 0, SEEK_END);
    const long long bytes = std::ftell(f);
    if (bytes < 0 || (unsigned long long)(bytes - (long long)sizeof(h)) != h.n * 24ull) {
        std::fclose(f);
        return pts;
    }
    std::fseek(f, (long)sizeof(h), SEEK_SET);
    pts.reserve((size_t)h.n);
    std::vector<double> buf;
    const size_t chunk = 1 << 20;
    for (size_t b = 0; b < (size_t)h.n; b += chunk) {
        const size_t e = std::min((size_t)h.n, b + chunk);
        buf.resize((e - b) * 3);
        if (std::fread(buf.data(), sizeof(double), buf.size(), f) != buf.size()) {
            pts.clear();
            break;
        }
        for (size_t i = 0; i < e - b; ++i) pts.emplace_back(buf[3 * i], buf[3 * i + 1], buf[3 * i + 2]);
    }
    std::fclose(f);
    if (dim) *dim = (int)h.dim;
    return pts;
}
