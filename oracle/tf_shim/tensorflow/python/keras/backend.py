from tensorflow.keras.backend import *  # noqa: F401,F403
from tensorflow.keras.backend import batch_get_value, batch_set_value, clear_session, floatx  # noqa: F401
