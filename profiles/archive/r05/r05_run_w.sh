#!/bin/bash
# round 5: is it the stealing?  w2 with the steal compiled out (ns), with thieves that only take from a view with > 256 tiles left (s256), w2, vxw -- same box
bash profiles/ab_run.sh r05w "vxw ns s256 w2" 3 -
