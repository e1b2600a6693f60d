// conv3p.hip — the bf16 position-tile kernel behind mv2_conv3 (declarations shared with conv3.hip, which owns the C ABI entry points)
#pragma once
#include "conv_common.h"

bool conv3p_eligible(const mv2_conv_desc& d);                 // shapes the kernel takes (checked after conv3.hip's eligible3)
size_t conv3p_workspace_bytes(const mv2_conv_desc& d);        // fp32 slabs of the tiles split across workgroups (0: no tile is split)
int conv3p_launch(const mv2_conv_desc* d, const void* in, size_t in_bytes, const void* wp3, const float* bias,
                  const void* wp2, const float* bias2, int act2, const float* se_wk, float se_bk, float* se_part, void* out,
                  void* workspace, size_t workspace_bytes, uint32_t* counters, size_t counter_bytes, hipStream_t st);
