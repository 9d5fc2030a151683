#pragma once
#include "../reference_types_min.hpp"
