#pragma once
#include "sequence_traits.h"
