// compat/nvbio/basic/thrust_view.h -- plain / device views of thrust vectors (nvbio/basic/thrust_view.h): they live in vector.h here.
#pragma once
#include "vector.h"
