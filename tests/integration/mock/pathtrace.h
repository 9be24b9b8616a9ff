// Stand-in for the reference's Inference/src/pathtrace.h:6-8 (the three prototypes the hot path is called through).
#pragma once
#include "sceneStructs.h"
struct uchar4 { unsigned char x, y, z, w; };
void pathtraceInit(Scene* scene);
void pathtraceFree();
void pathtrace(uchar4* pbo, int frame, int iteration);
