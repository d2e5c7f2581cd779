// vo/vo.hpp -- umbrella header of the C++ facade.
#pragma once
#include "camera.hpp"
#include "context.hpp"
#include "epipolar.hpp"
#include "evaluation.hpp"
#include "kdtree.hpp"
#include "files.hpp"
#include "picp_solver.hpp"
#include "point_cloud.hpp"
#include "sequence.hpp"
#include "types.hpp"
#include "utils.hpp"
