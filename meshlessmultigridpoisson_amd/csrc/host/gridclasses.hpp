// gridclasses.hpp -- configuration / boundary types of the reference's class
// surface (MeshlessPoisson/gridclasses.hpp:6-28), same member names.
#ifndef MMGH_GRID_CLASSES_H
#define MMGH_GRID_CLASSES_H
#include <vector>

#include "la.hpp"

class GridProperties {  // gridclasses.hpp:6-14
public:
    int rbfExp = 3;
    int polyDeg = 3;
    int laplaceMatSize = 0;
    int stencilSize = 25;
    double omega = 1.4;
    int iters = 5;
};

class Boundary {  // gridclasses.hpp:15-20
public:
    int type = 0;  // 1 dirichlet, 2 neumann (set by Grid::setBCFlag)
    std::vector<int> bcPoints;
    std::vector<double> values;
};

class deriv_normal_bc {  // gridclasses.hpp:21-28
public:
    int pointID = 0;
    mmgh::Vec weights;
    std::vector<int> neighbors;
    double value = 0;
};
#endif
