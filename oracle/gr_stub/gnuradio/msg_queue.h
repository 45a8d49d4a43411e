#pragma once
#include <gnuradio/stub_runtime.h>
