#pragma once
#include "alignment.h"
