#!/bin/bash
python tools/late_step_profile.py 2>&1 | grep -v amdgpu
