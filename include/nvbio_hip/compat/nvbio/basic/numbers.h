#pragma once
#include "types.h"
