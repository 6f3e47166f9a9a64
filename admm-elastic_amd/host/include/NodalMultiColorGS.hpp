#include "LinearSolver.hpp"
