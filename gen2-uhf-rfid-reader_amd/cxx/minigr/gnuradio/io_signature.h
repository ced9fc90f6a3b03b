// minigr: gr::io_signature lives in block.h here
#pragma once
#include <gnuradio/block.h>
