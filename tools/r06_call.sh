#!/bin/bash
# round 6: the profile-mode test with its loosened timing bounds, three times
for i in 1 2 3; do timeout 300 python -m pytest tests/test_hip_profile.py -x -q -m gpu 2>&1 | tail -1; done
