// Part of the seal:: facade over libmm_hip (see seal/seal.h).
#pragma once
#include "seal/seal.h"
