"""transvae -- MI355X-native TransVAE forward/backward path.

Import surface of the reference package (R/transvae/__init__.py:5-9): `from transvae import
TransVAE, create_transvae`.  (The reference also re-exports its loss, which needs the `lpips`
package and external VGG weights; the loss is a consumer of this path and out of scope here.)
"""
from .models.transvae import TransVAE, create_transvae

__version__ = "0.1.0"
__all__ = ["TransVAE", "create_transvae"]
