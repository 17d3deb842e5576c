#pragma once
#include "sequence.h"
