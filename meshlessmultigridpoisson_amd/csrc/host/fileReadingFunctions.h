// fileReadingFunctions.h -- Gmsh MSH 2.x ASCII / txt readers and the txt writer
// with the reference's names (MeshlessPoisson/fileReadingFunctions.h:10-14).
// Bit-compatible on well-formed input; unlike the reference a missing file gives
// an empty result instead of a crash on a NULL FILE*.
#ifndef MMGH_FILE_READING_FUNCTIONS_H
#define MMGH_FILE_READING_FUNCTIONS_H
#include <tuple>
#include <utility>
#include <vector>

std::vector<std::tuple<double, double, double>> pointsFromMshFile(const char *fname);
std::vector<std::tuple<double, double, double>> pointsFromTxts(const char *fname);
std::vector<std::pair<int, int>> boundPtsConnFromMsh(const char *fname, const std::vector<int> &bcFlags);
std::vector<int> orderFromTxt(const char *fname, int nv);
void writeVectorToTxt(std::vector<double> vec, const char *filename);
// Not in the reference: MSH 2.2 ASCII $Nodes writer for synthetic clouds
// (plus optional triangle elements, 1-based node ids).
bool writePointsToMshFile(const char *fname, const std::vector<std::tuple<double, double, double>> &pts,
                          const std::vector<int> *triangles = nullptr);
// Not in the reference: binary point-cloud container for 1e7+ points.  An MSH 2.2
// ASCII file of 1e7 nodes is ~0.7 GB of text and minutes of fscanf; this is 24 B per point read
// in one pass.  Layout (little-endian): char magic[8] = "MMGCLOUD", u32 version = 1, u32 dim,
// u64 n, then n records of 3 doubles (x, y, z; z = 0 when dim == 2).  Readers return an empty
// vector on a missing / truncated / foreign file, like the text readers above.
bool writePointsToBinFile(const char *fname, const std::vector<std::tuple<double, double, double>> &pts, int dim);
std::vector<std::tuple<double, double, double>> pointsFromBinFile(const char *fname, int *dim = nullptr);
#endif
