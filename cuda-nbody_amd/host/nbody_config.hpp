// nbody_config.hpp -- initial-condition selector, values as in /root/reference/src/nbody/nbody_config.hpp:3.
#pragma once

enum class NBodyConfig { NBODY_CONFIG_RANDOM, NBODY_CONFIG_SHELL, NBODY_CONFIG_EXPAND, NBODY_NUM_CONFIGS };
