#!/bin/bash
# round 5: what would a polyphase stage that keeps its taps in registers save at most?  Diagnostic builds of k_front_mid (timing only,
# wrong bytes): lean = nothing fetched a phase ahead (126 VGPRs), leanng = lean + the tap gather only once per run (168 VGPRs, 22 spills),
# nogather = the shipped order + the gather once per run (168 VGPRs, 63 spills).  Same box, alternating (tools/abn.sh).
cd $GRAFT_REPO_ROOT
bash tools/abn.sh new lean leanng nogather
