// Part of the seal:: facade over libmm_hip (see seal/seal.h): SEAL splits its API over many headers, the reference
// includes a few of them by name; every one of them is the whole facade here.
#pragma once
#include "seal/seal.h"
