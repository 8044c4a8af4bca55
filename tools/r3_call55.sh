#!/bin/bash
# round 3 call 55: fp16 hi pass + block-scaled e4m3 lo pass on the device (prototype: conversion on the device, weights packed on the host)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
(timeout 20 tools/bin/mx_lo_pass_proto 768; timeout 20 tools/bin/mx_lo_pass_proto 2816) > $O/mx_lo_pass_proto.jsonl 2>&1; echo "proto rc=$?" > $O/rc.txt
cat $O/rc.txt $O/mx_lo_pass_proto.jsonl
