#!/bin/bash
cd /root/repo/motioneditor_amd
cp libmotioned.so /tmp/new.so; cp libmotioned_old.so /tmp/old.so
cd /root/repo
for v in old new old new; do
cp /tmp/$v.so motioneditor_amd/libmotioned.so
echo $v; timeout 300 python tools/kbench.py attn 2>&1 | grep "L0 prev\|L0 edited\|L1 edited\|L2"
done
