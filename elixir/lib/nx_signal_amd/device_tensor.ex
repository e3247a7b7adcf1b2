defmodule NxSignalAMD.DeviceTensor do
  @moduledoc """
  A tensor that lives in the GPU's HBM: a NIF resource (freed by its destructor when the last reference goes) plus the
  shape / type / names an `Nx.Tensor` would carry.  It is what keeps the reference's documented STFT-domain filtering
  chain (`guides/filtering.livemd:137-159`: `stft -> Nx.multiply(z, hfft) -> istft`) on the device instead of crossing
  PCIe twice per step:

      x   = NxSignalAMD.DeviceTensor.to_device(signal)
      {z, t, f} = NxSignalAMD.stft(x, window, overlap_length: 768, sampling_rate: 48_000)   # z stays in HBM
      z   = NxSignalAMD.DeviceTensor.multiply_spectrum(z, hfft)                             # in place
      y   = NxSignalAMD.istft(z, window, overlap_length: 768, sampling_rate: 48_000)        # still in HBM
      out = NxSignalAMD.DeviceTensor.from_device(y)                                         # one download

  Device calls are asynchronous on the context's HIP stream; `from_device/1` and `sync/1` wait.
  """
  alias NxSignalAMD.NIF

  @enforce_keys [:ref, :ctx, :shape, :type]
  defstruct [:ref, :ctx, :shape, :type, names: nil]

  @type t :: %__MODULE__{ref: reference(), ctx: reference(), shape: tuple(), type: {:f, 32} | {:c, 64}, names: list() | nil}

  @doc "Uploads an `Nx.Tensor` (f32 or c64; other real types are converted to f32 like `NxSignal` does for `Nx.fft`)."
  def to_device(%Nx.Tensor{} = tensor, opts \\ []) do
    opts = Keyword.validate!(opts, device: 0)
    ctx = NxSignalAMD.context(opts[:device])

    {tensor, type} =
      case Nx.type(tensor) do
        {:c, 64} -> {tensor, {:c, 64}}
        {:c, _} -> raise ArgumentError, "only c64 complex tensors are supported, got: #{inspect(Nx.type(tensor))}"
        {:f, 64} -> raise ArgumentError, "f64 tensors are not supported by the MI355X path (the reference computes them in double)"
        _ -> {Nx.as_type(tensor, :f32), {:f, 32}}
      end

    {:ok, ref} = NIF.to_device(ctx, Nx.to_binary(tensor)) |> NxSignalAMD.unwrap!()
    %__MODULE__{ref: ref, ctx: ctx, shape: Nx.shape(tensor), type: type, names: Nx.names(tensor)}
  end

  @doc "Downloads into an `Nx.Tensor` (waits for the stream)."
  def from_device(%__MODULE__{} = dt) do
    {:ok, bin} = NIF.from_device(dt.ref) |> NxSignalAMD.unwrap!()
    t = Nx.from_binary(bin, dt.type) |> Nx.reshape(dt.shape)
    if dt.names, do: Nx.rename(t, dt.names), else: t
  end

  @doc "Waits for everything enqueued on the tensor's context."
  def sync(%__MODULE__{ctx: ctx}), do: NIF.sync(ctx) |> NxSignalAMD.unwrap!()

  @doc """
  `Nx.multiply(z, h)` for a device spectrum `z :: c64[..., frames, K]` and a host `h :: c64[K]` (the DFT of a filter),
  in place, each component rounded once like `Nx.BinaryBackend`'s complex multiply.
  """
  def multiply_spectrum(%__MODULE__{type: {:c, 64}} = z, %Nx.Tensor{} = h) do
    k = elem(z.shape, tuple_size(z.shape) - 1)

    if Nx.shape(h) != {k} do
      raise ArgumentError, "expected a filter spectrum of shape {#{k}}, got: #{inspect(Nx.shape(h))}"
    end

    rows = div(Tuple.product(z.shape), k)
    hb = h |> Nx.as_type(:c64) |> Nx.to_binary()
    {:ok, _} = NIF.spectrum_mul_dev(z.ctx, z.ref, rows, k, hb) |> NxSignalAMD.unwrap!()
    z
  end
end
