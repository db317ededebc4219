// feat.h -- compiled feature graph (aasr_feat); see feat_graph.cc.
#pragma once
#include "common.h"
